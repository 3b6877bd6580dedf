/*
 * animation.hpp -- transformations over time: the Animation interface, key frame animations and the
 * per-time cache of their matrices (reference: animation.hpp:38-127, animation_keyframes.hpp:51-214).
 *
 * On the host these serve the scene description and the bounding boxes of moving hitables
 * (Scene::updateBVH(t0, t1)); at render time the kernels evaluate the same key frames at each ray's
 * own time (wurblpt_amd/csrc/wpt_anim.h), so only AnimationKeyframes can go to the device.
 */
#pragma once

#include <algorithm>
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
    std::vector<Keyframe> _keyframes; /* strictly ascending in time */

    /* first key frame that is not earlier than t */
    std::vector<Keyframe>::const_iterator notBefore(float t) const
    {
        return std::lower_bound(_keyframes.begin(), _keyframes.end(), t, [](const Keyframe& k, float time) { return k.t < time; });
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

    /* keeps the list sorted; a key frame with the same time stamp is replaced */
    void addKeyframe(const Keyframe& keyframe)
    {
        auto where = notBefore(keyframe.t);
        const size_t index = where - _keyframes.begin();
        if (where != _keyframes.end() && where->t == keyframe.t)
            _keyframes[index] = keyframe;
        else
            _keyframes.insert(_keyframes.begin() + index, keyframe);
    }
    void addKeyframe(float time, const Transformation& transf) { addKeyframe(Keyframe(time, transf)); }

    float startTime() const { return _keyframes.empty() ? 0.0f : _keyframes.front().t; }
    float endTime() const { return _keyframes.empty() ? 0.0f : _keyframes.back().t; }

    /* the first / last key frame outside their range, a key frame at its own time, mix() of the neighbours between */
    virtual Transformation at(float t) const override
    {
        if (_keyframes.empty())
            return Transformation();
        if (t <= startTime())
            return _keyframes.front().transformation;
        if (t >= endTime())
            return _keyframes.back().transformation;
        auto higher = notBefore(t);
        if (higher->t == t)
            return higher->transformation;
        auto lower = higher - 1;
        const float alpha = 1.0f - (higher->t - t) / (higher->t - lower->t);
        return mix(lower->transformation, higher->transformation, alpha);
    }
};

/* The transformations of a scene's animations at one moment, each with its matrix and normal matrix; an entry is
 * evaluated when it is first asked for (reference: AnimationCache, animation.hpp:52-125) */
class AnimationCache
{
private:
    struct Entry {
        Transformation transformation;
        mat4 M;
        mat3 N;
        bool valid = false;
    };
    const std::vector<const Animation*>* _animations = nullptr;
    float _t = std::numeric_limits<float>::max();
    std::vector<Entry> _entries;

    const Entry& entry(int animationIndex)
    {
        Entry& e = _entries[animationIndex];
        if (!e.valid) {
            e.transformation = (*_animations)[animationIndex]->at(_t);
            e.M = e.transformation.toMat4();
            e.N = e.transformation.toNormalMatrix();
            e.valid = true;
        }
        return e;
    }

public:
    AnimationCache() {}
    AnimationCache(const std::vector<const Animation*>& animations) : _animations(&animations), _entries(animations.size()) {}
    /* every animation evaluated at t right away */
    AnimationCache(const std::vector<const Animation*>& animations, float t) : AnimationCache(animations)
    {
        _t = t;
        for (size_t i = 0; i < _entries.size(); i++)
            entry(int(i));
    }
    /* another moment: all entries are stale */
    void init(float t)
    {
        _t = t;
        for (Entry& e : _entries)
            e.valid = false;
    }
    const Transformation& get(int animationIndex) { return entry(animationIndex).transformation; }
    const mat4& getM(int animationIndex) { return entry(animationIndex).M; }
    const mat3& getN(int animationIndex) { return entry(animationIndex).N; }
    float t() const { return _t; }
};

}
