/*
 * tgd/array.hpp -- this framework's own 2D array container, published under the include path and in the namespace the
 * reference's applications use for libtgd's (`#include <tgd/array.hpp>`, `TGD::Array<float>`;
 * wurblpt-cornellbox.cpp:26,271-273), so that they compile against include/ unchanged where libtgd itself is not
 * installed.  It is NOT libtgd: only what libwurblpt's public interface and those applications touch is here -- a
 * 2D array of interleaved components with libtgd's accessor names (texture_image.hpp:45, sensor_rgb.hpp:37), x
 * fastest, then y; copies share the pixel storage.  With libtgd installed and in front of this directory in the
 * include path the real container is taken instead (untested here: the library is not in this image).
 */
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace TGD {

enum ComponentType { uint8 = 0, uint16 = 1, float32 = 2, int32 = 3 };

inline size_t componentTypeSize(ComponentType t) { return t == uint8 ? 1 : t == uint16 ? 2 : 4; }

class TagList
{
private:
    std::map<std::string, std::string> _tags;

public:
    void set(const std::string& key, const std::string& value) { _tags[key] = value; }
    std::string value(const std::string& key, const std::string& def = std::string()) const
    {
        auto it = _tags.find(key);
        return it == _tags.end() ? def : it->second;
    }
    bool contains(const std::string& key) const { return _tags.find(key) != _tags.end(); }
};

class ArrayContainer
{
private:
    size_t _dims[2];
    size_t _comps;
    ComponentType _type;
    std::shared_ptr<std::vector<unsigned char>> _data;
    std::shared_ptr<TagList> _globalTags;

public:
    ArrayContainer() : _dims { 0, 0 }, _comps(0), _type(uint8), _globalTags(new TagList) {}
    ArrayContainer(size_t width, size_t height, size_t comps, ComponentType type) :
        _dims { width, height }, _comps(comps), _type(type),
        _data(new std::vector<unsigned char>(width * height * comps * componentTypeSize(type), 0)),
        _globalTags(new TagList)
    {
    }

    size_t dimensionCount() const { return 2; }
    size_t dimension(size_t i) const { return _dims[i]; }
    size_t componentCount() const { return _comps; }
    ComponentType componentType() const { return _type; }
    size_t componentSize() const { return componentTypeSize(_type); }
    size_t elementCount() const { return _dims[0] * _dims[1]; }
    size_t elementSize() const { return _comps * componentSize(); }
    size_t dataSize() const { return elementCount() * elementSize(); }
    void* data() { return _data ? _data->data() : nullptr; }
    const void* data() const { return _data ? _data->data() : nullptr; }
    TagList& globalTagList() { return *_globalTags; }
    const TagList& globalTagList() const { return *_globalTags; }

    template<typename T> T* get(size_t elementIndex)
    {
        return reinterpret_cast<T*>(_data->data() + elementIndex * elementSize());
    }
    template<typename T> const T* get(size_t elementIndex) const
    {
        return reinterpret_cast<const T*>(_data->data() + elementIndex * elementSize());
    }
    template<typename T> T* get(size_t x, size_t y) { return get<T>(y * _dims[0] + x); }
    template<typename T> const T* get(size_t x, size_t y) const { return get<T>(y * _dims[0] + x); }
};

template<typename T> struct ComponentTypeOf;
template<> struct ComponentTypeOf<uint8_t> { static constexpr ComponentType value = uint8; };
template<> struct ComponentTypeOf<uint16_t> { static constexpr ComponentType value = uint16; };
template<> struct ComponentTypeOf<float> { static constexpr ComponentType value = float32; };
template<> struct ComponentTypeOf<int32_t> { static constexpr ComponentType value = int32; };

template<typename T> class Array : public ArrayContainer
{
public:
    Array() {}
    Array(size_t width, size_t height, size_t comps) : ArrayContainer(width, height, comps, ComponentTypeOf<T>::value) {}
    /* a container whose components already have this type (shares its storage); anything else is an empty array */
    Array(const ArrayContainer& container) : ArrayContainer(container.componentType() == ComponentTypeOf<T>::value ? container : ArrayContainer()) {}
    T* operator[](size_t elementIndex) { return this->template get<T>(elementIndex); }
    const T* operator[](size_t elementIndex) const { return this->template get<T>(elementIndex); }
    T* at(size_t x, size_t y) { return this->template get<T>(x, y); }
    const T* at(size_t x, size_t y) const { return this->template get<T>(x, y); }
};

}
