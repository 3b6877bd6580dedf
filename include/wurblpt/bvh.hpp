/*
 * bvh.hpp -- host-side BVH construction; the node ORDER is part of the results contract.
 *
 * The kernel visits nodes in the reference's order (unordered depth-first, left child first;
 * ties in hit distance are won by the later candidate), so the tree must be the tree the
 * reference builds (bvh.hpp:93-193): top-down, full-sweep SAH on the longest axis of the
 * node's box after std::sort of the subset by box centre, one hitable per leaf, flattened
 * depth-first into 32-byte nodes (bvh.hpp:217-246).  Large subtrees are built by two OpenMP
 * sections exactly where the reference does; the ranges are disjoint, so the result does not
 * depend on the thread count.
 */
#pragma once

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <omp.h>

#include "../wurblpt_hip.h"
#include "gvm.hpp"

namespace WurblPT {

class AABB
{
public:
    vec3 lo, hi;
    AABB() {}
    AABB(const vec3& lo, const vec3& hi) : lo(lo), hi(hi) {}
    vec3 center() const { return 0.5f * (lo + hi); }
    int longestAxis() const
    {
        int axis = 2;
        vec3 l = hi - lo;
        if (l[0] > l[1] && l[0] > l[2])
            axis = 0;
        else if (l[1] > l[2])
            axis = 1;
        return axis;
    }
    float surfaceArea() const
    {
        vec3 l = hi - lo;
        return 2.0f * (l.x() * l.y() + l.y() * l.z() + l.x() * l.z());
    }
};

inline AABB merge(const AABB& a, const AABB& b) { return AABB(min(a.lo, b.lo), max(a.hi, b.hi)); }

class BVHBuilder
{
private:
    static constexpr size_t maxTreeDepth = 128; /* traversal stack size in the kernel, bvh.hpp:230 */
    const std::vector<AABB>& _boxes;
    std::vector<unsigned int> _subset;
    std::vector<float> _areas0, _areas1;

    struct Node {
        AABB box;
        Node* child[2] = { nullptr, nullptr };
        unsigned int primitive = 0;
        ~Node()
        {
            delete child[0];
            delete child[1];
        }
    };

    void buildNode(Node* node, size_t offset, size_t count)
    {
        if (count == 1) {
            node->box = _boxes[_subset[offset]];
            node->primitive = _subset[offset];
            return;
        }
        int n = count;
        AABB box = _boxes[_subset[offset]];
        for (int i = 1; i < n; i++)
            box = merge(box, _boxes[_subset[offset + i]]);
        node->box = box;
        const int axis = box.longestAxis();
        const std::vector<AABB>& boxes = _boxes;
        std::sort(_subset.begin() + offset, _subset.begin() + offset + count,
                [&boxes, axis](unsigned int i, unsigned int j) { return boxes[i].center()[axis] < boxes[j].center()[axis]; });
        /* sweep: area of the union of the first i+1 boxes, and of the boxes from i to the end */
        AABB left = _boxes[_subset[offset]];
        _areas0[offset] = left.surfaceArea();
        for (int i = 1; i < n - 1; i++) {
            left = merge(left, _boxes[_subset[offset + i]]);
            _areas0[offset + i] = left.surfaceArea();
        }
        AABB right = _boxes[_subset[offset + n - 1]];
        _areas1[offset + n - 1] = right.surfaceArea();
        for (int i = n - 2; i > 0; i--) {
            right = merge(right, _boxes[_subset[offset + i]]);
            _areas1[offset + i] = right.surfaceArea();
        }
        auto cost = [this, offset, count](int i) { return i * _areas0[offset + i - 1] + (count - i) * _areas1[offset + i]; };
        int split = 1;
        float best = cost(1);
        for (int i = 2; i < n; i++) {
            float c = cost(i);
            if (c < best) {
                best = c;
                split = i;
            }
        }
        size_t count0 = split, count1 = count - count0;
        node->child[0] = new Node;
        node->child[1] = new Node;
        constexpr unsigned int parallelizationThreshold = 16384;
        if (count0 >= parallelizationThreshold && count1 >= parallelizationThreshold) {
#pragma omp parallel sections num_threads(2)
            {
#pragma omp section
                buildNode(node->child[0], offset, count0);
#pragma omp section
                buildNode(node->child[1], offset + split, count1);
            }
        } else {
            buildNode(node->child[0], offset, count0);
            buildNode(node->child[1], offset + split, count1);
        }
    }

    static void measure(const Node* node, size_t& nodeCount, size_t& depthMax, size_t depth)
    {
        nodeCount++;
        if (depth > depthMax)
            depthMax = depth;
        if (node->child[0]) {
            measure(node->child[0], nodeCount, depthMax, depth + 1);
            measure(node->child[1], nodeCount, depthMax, depth + 1);
        }
    }

    static uint32_t flatten(const Node* node, std::vector<wpt_bvh_node>& out, uint32_t primitiveKind)
    {
        uint32_t me = uint32_t(out.size());
        out.push_back(wpt_bvh_node());
        for (int k = 0; k < 3; k++) {
            out[me].lo[k] = node->box.lo[k];
            out[me].hi[k] = node->box.hi[k];
        }
        if (!node->child[0]) {
            out[me].kind = primitiveKind;
            out[me].link = node->primitive;
        } else {
            flatten(node->child[0], out, primitiveKind);
            uint32_t second = flatten(node->child[1], out, primitiveKind);
            out[me].kind = WPT_NODE_INNER;
            out[me].link = second;
        }
        return me;
    }

public:
    explicit BVHBuilder(const std::vector<AABB>& boxes) : _boxes(boxes) {}

    /* Returns the linearized tree; `levels` receives the depth. */
    std::vector<wpt_bvh_node> build(size_t* levels = nullptr)
    {
        std::vector<wpt_bvh_node> out;
        if (_boxes.size() == 0) {
            /* the reference's empty scene: one leaf with a zero box and no hitable (bvh.hpp:188-191) */
            wpt_bvh_node n;
            memset(&n, 0, sizeof(n));
            n.kind = WPT_NODE_EMPTY;
            out.push_back(n);
            if (levels)
                *levels = 1;
            return out;
        }
        _subset.resize(_boxes.size());
        for (size_t i = 0; i < _boxes.size(); i++)
            _subset[i] = i;
        _areas0.resize(_boxes.size());
        _areas1.resize(_boxes.size());
        Node* root = new Node;
        int activeLevels = omp_get_max_active_levels();
        omp_set_max_active_levels(maxTreeDepth);
        buildNode(root, 0, _subset.size());
        omp_set_max_active_levels(activeLevels);
        size_t nodeCount = 0, depthMax = 0;
        measure(root, nodeCount, depthMax, 1);
        if (depthMax > maxTreeDepth) {
            fprintf(stderr, "BVH: too many levels (%zu)\n", depthMax);
            abort();
        }
        out.reserve(nodeCount);
        flatten(root, out, WPT_NODE_TRIANGLE);
        delete root;
        if (levels)
            *levels = depthMax;
        return out;
    }
};

}
