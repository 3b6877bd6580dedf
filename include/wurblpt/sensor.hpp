/*
 * sensor.hpp -- the film (reference sensor.hpp:36-82, sensor_rgb.hpp:35-98).
 * accumulateRadiance()/finishPixel() run in the HIP kernel; SensorRGB holds the frame,
 * float[height][width][3] with row 0 at the bottom, and the four gate values.
 */
#pragma once

#include <limits>

#include "array.hpp"

namespace WurblPT {

class Sensor
{
public:
    constexpr static int maxPixelComponents = 3;
    virtual ~Sensor() {}
    virtual unsigned int width() const { return 0; }
    virtual unsigned int height() const { return 0; }
    virtual ArrayContainer* pixelArray() { return nullptr; }
    virtual float aspectRatio() const { return float(width()) / height(); }
};

class SensorRGB final : public Sensor
{
private:
    Array<float> _frame;

public:
    const float minDistToLight, maxDistToLight;
    const float minPathLen, maxPathLen;

    SensorRGB(unsigned int width, unsigned int height, float minDistToLight = 0.0f,
            float maxDistToLight = std::numeric_limits<float>::max(), float minPathLen = 0.0f,
            float maxPathLen = std::numeric_limits<float>::max()) :
        _frame(width, height, 3), minDistToLight(minDistToLight), maxDistToLight(maxDistToLight), minPathLen(minPathLen),
        maxPathLen(maxPathLen)
    {
    }
    virtual unsigned int width() const override { return _frame.dimension(0); }
    virtual unsigned int height() const override { return _frame.dimension(1); }
    virtual ArrayContainer* pixelArray() override { return &_frame; }
    const Array<float>& result() const { return _frame; }
};

}
