/*
 * animation.hpp -- transformations over time: the Animation interface, key frame animations and the
 * per-time cache of their matrices (reference: animation.hpp:38-127, animation_keyframes.hpp:51-214).
 *
 * On the host these serve the scene description and the bounding boxes of moving hitables
 * (Scene::updateBVH(t0, t1)); at render time the kernels evaluate the same key frames at each ray's
 * own time (wurblpt_amd/csrc/wpt_anim.h), so only AnimationKeyframes can go to the device.
 */
#pragma once

#include <limits>
#include <vector>

#include "transformation.hpp"

namespace WurblPT {

class Animation
{
public:
    Animation() {}
    virtual ~Animation() {}
    virtual Transformation at(float t) const = 0;
};

class AnimationKeyframes : public Animation
{
public:
    class Keyframe
    {
    public:
        float t;                       /* seconds */
        Transformation transformation; /* of the target at time t */
        Keyframe() : t(0), transformation() {}
        Keyframe(float time, const Transformation& transf) : t(time), transformation(transf) {}
    };

private:
    std::vector<Keyframe> _keyframes; /* ascending time */

    /* the neighbours of t among the key frames; both indices are equal on an exact match */
    void findKeyframeIndices(float t, int& lowerIndex, int& higherIndex) const
    {
        int a = 0;
        int b = int(_keyframes.size()) - 1;
        while (b >= a) {
            int c = (a + b) / 2;
            if (_keyframes[c].t < t) {
                a = c + 1;
            } else if (_keyframes[c].t > t) {
                b = c - 1;
            } else {
                lowerIndex = higherIndex = c;
                return;
            }
        }
        lowerIndex = b;
        higherIndex = a;
    }

public:
    AnimationKeyframes() {}
    explicit AnimationKeyframes(float t0, const Transformation& T0, float t1, const Transformation& T1)
    {
        addKeyframe(t0, T0);
        addKeyframe(t1, T1);
    }
    explicit AnimationKeyframes(const std::vector<Keyframe>& keyframes) : _keyframes(keyframes) {}

    const std::vector<Keyframe>& keyframes() const { return _keyframes; }

    /* a key frame with the same time stamp is replaced */
    void addKeyframe(const Keyframe& keyframe)
    {
        if (_keyframes.empty() || keyframe.t > endTime()) {
            _keyframes.push_back(keyframe);
        } else if (keyframe.t < startTime()) {
            _keyframes.insert(_keyframes.begin(), keyframe);
        } else {
            int lowerIndex, higherIndex;
            findKeyframeIndices(keyframe.t, lowerIndex, higherIndex);
            if (lowerIndex == higherIndex)
                _keyframes[lowerIndex] = keyframe;
            else
                _keyframes.insert(_keyframes.begin() + higherIndex, keyframe);
        }
    }
    void addKeyframe(float time, const Transformation& transf) { addKeyframe(Keyframe(time, transf)); }

    float startTime() const { return _keyframes.empty() ? 0.0f : _keyframes.front().t; }
    float endTime() const { return _keyframes.empty() ? 0.0f : _keyframes.back().t; }

    virtual Transformation at(float t) const override
    {
        if (_keyframes.empty())
            return Transformation();
        if (t <= startTime())
            return _keyframes.front().transformation;
        if (t >= endTime())
            return _keyframes.back().transformation;
        int lowerIndex, higherIndex;
        findKeyframeIndices(t, lowerIndex, higherIndex);
        if (lowerIndex == higherIndex)
            return _keyframes[lowerIndex].transformation;
        float alpha = 1.0f - (_keyframes[higherIndex].t - t) / (_keyframes[higherIndex].t - _keyframes[lowerIndex].t);
        return mix(_keyframes[lowerIndex].transformation, _keyframes[higherIndex].transformation, alpha);
    }
};

/* The transformations of all animations of a scene at one time, with their matrices */
class AnimationCache
{
private:
    const std::vector<const Animation*>* _animations;
    float _t;
    std::vector<Transformation> _transformations;
    std::vector<mat4> _transformationMs;
    std::vector<mat3> _transformationNs;
    std::vector<bool> _initialized;

    void initIndexIfNecessary(int i)
    {
        if (!_initialized[i]) {
            _transformations[i] = (*_animations)[i]->at(_t);
            _transformationMs[i] = _transformations[i].toMat4();
            _transformationNs[i] = _transformations[i].toNormalMatrix();
            _initialized[i] = true;
        }
    }

public:
    AnimationCache() : _animations(nullptr), _t(std::numeric_limits<float>::max()) {}
    AnimationCache(const std::vector<const Animation*>& animations) :
        _animations(&animations), _t(std::numeric_limits<float>::max()), _transformations(animations.size()),
        _transformationMs(animations.size()), _transformationNs(animations.size()), _initialized(animations.size(), false)
    {
    }
    AnimationCache(const std::vector<const Animation*>& animations, float t) : AnimationCache(animations)
    {
        _t = t;
        for (size_t i = 0; i < _animations->size(); i++)
            initIndexIfNecessary(i);
    }
    void init(float t)
    {
        _t = t;
        for (size_t i = 0; i < _initialized.size(); i++)
            _initialized[i] = false;
    }
    const Transformation& get(int animationIndex)
    {
        initIndexIfNecessary(animationIndex);
        return _transformations[animationIndex];
    }
    const mat4& getM(int animationIndex)
    {
        initIndexIfNecessary(animationIndex);
        return _transformationMs[animationIndex];
    }
    const mat3& getN(int animationIndex)
    {
        initIndexIfNecessary(animationIndex);
        return _transformationNs[animationIndex];
    }
    float t() const { return _t; }
};

}
