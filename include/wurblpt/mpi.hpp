/*
 * mpi.hpp -- block hand-out with MPICoordinator semantics (reference mpi.hpp:152-289).
 *
 * The reference distributes blocks of consecutive pixel indices to MPI ranks through a
 * coordinator thread on rank 0 (mpi.hpp:69-107).  Inside one node the same semantics need no
 * messages: worker threads (one per GPU) pull block indices from one atomic counter
 * (getBlock) and copy finished blocks straight into the final frame (submitBlock).  With one
 * device the whole frame is one block, as in the reference's single-process path (:241-254).
 */
#pragma once

#include <atomic>
#include <cstdio>
#include <vector>

namespace WurblPT {

class MPICoordinator
{
private:
    unsigned int _blockSize;
    std::vector<int> _devices;
    std::atomic<unsigned int> _nextBlock;
    unsigned int _pixelCount;
    unsigned int _effectiveBlockSize;
    float* _pixelData;
    unsigned int _componentCount;

public:
    /* blockSize as in the reference (default 4096 pixels) is a lower bound here: a GPU needs
     * far more pixels in flight than a CPU rank, so with several devices the frame is cut into
     * about 4 blocks per device, never smaller than blockSize. */
    MPICoordinator(unsigned int blockSize = 4096, const std::vector<int>& devices = std::vector<int>()) :
        _blockSize(blockSize), _devices(devices), _nextBlock(0), _pixelCount(0), _effectiveBlockSize(0), _pixelData(nullptr),
        _componentCount(0)
    {
        if (_devices.empty())
            _devices.push_back(-1); /* the current device */
    }

    const char* processId() const { return "main"; }
    const std::vector<int>& devices() const { return _devices; }

    void init(unsigned int width, unsigned int height, float* pixelData, unsigned int componentCount)
    {
        _pixelCount = width * height;
        _pixelData = pixelData;
        _componentCount = componentCount;
        _nextBlock = 0;
        if (_devices.size() == 1) {
            _effectiveBlockSize = _pixelCount;
        } else {
            unsigned int target = _pixelCount / (4 * _devices.size());
            _effectiveBlockSize = target > _blockSize ? target : _blockSize;
        }
    }

    /* thread safe */
    void getBlock(unsigned int* blockStart, unsigned int* blockSize)
    {
        unsigned int index = _nextBlock.fetch_add(1);
        unsigned long long start = (unsigned long long)(index) * _effectiveBlockSize;
        if (_effectiveBlockSize == 0 || start >= _pixelCount) {
            *blockStart = 0;
            *blockSize = 0;
        } else {
            *blockStart = start;
            *blockSize = (start + _effectiveBlockSize > _pixelCount) ? _pixelCount - start : _effectiveBlockSize;
        }
    }

    /* where a worker writes the block's pixels; the frame is shared, blocks are disjoint */
    float* blockData(unsigned int blockStart) const { return _pixelData + size_t(blockStart) * _componentCount; }
    void submitBlock(unsigned int /* blockStart */, unsigned int /* blockSize */) {}
    void finish() {}
    bool mainProcess() const { return true; }
    int worldSize() const { return 1; }
};

}
