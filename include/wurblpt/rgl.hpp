/*
 * rgl.hpp -- host side of the measured-BRDF material (MaterialRGL, material_rgl.hpp:46-102):
 * reads a BRDF file of the RGL material database (the tensor-file container,
 * powitacq_rgb.inl:728-803), builds the five interpolants / sample warps the model uses exactly
 * as its constructor does (powitacq_rgb.inl:213-310,893-1003) and appends them to the flat
 * float pool that the device reads.
 *
 * The model and the order of its operations are those of powitacq_rgb (Jonathan Dupuy and Wenzel Jakob, "An Adaptive
 * Parameterization for Efficient Material Acquisition and Rendering"), which is distributed under the 3-clause BSD
 * licence: Copyright 2018 Jonathan Dupuy and Wenzel Jakob.  Redistribution and use in source and binary forms, with or
 * without modification, are permitted provided that the conditions of that licence are met; its full text, with the
 * disclaimer, is reproduced in the LICENSE file of this repository.
 */
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../wurblpt_hip.h"

namespace WurblPT {

/* One field of a tensor file */
struct TensorField {
    unsigned int dtype = 0; /* 1 = uint8 ... 10 = float32 (powitacq_rgb.inl:652-664) */
    std::vector<size_t> shape;
    std::vector<unsigned char> bytes;
    const float* floats() const { return reinterpret_cast<const float*>(bytes.data()); }
    size_t count() const
    {
        size_t n = 1;
        for (size_t s : shape)
            n *= s;
        return n;
    }
};

/* Tensor file reader (powitacq_rgb.inl:728-803): "tensor_file\0", version 1.0, field table, data */
class TensorFile
{
public:
    std::map<std::string, TensorField> fields;
    std::string error;

    bool load(const std::string& filename)
    {
        FILE* f = fopen(filename.c_str(), "rb");
        if (!f) {
            error = "unable to open " + filename;
            return false;
        }
        std::vector<unsigned char> all;
        unsigned char buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof(buf), f)) > 0)
            all.insert(all.end(), buf, buf + got);
        fclose(f);
        auto bad = [&](const char* msg) {
            error = filename + ": " + msg;
            return false;
        };
        if (all.size() < 18 || memcmp(all.data(), "tensor_file", 12) != 0)
            return bad("not a tensor file");
        if (all[12] != 1 || all[13] != 0)
            return bad("unknown tensor file version");
        static const size_t typeSize[12] = { 0, 1, 1, 2, 2, 4, 4, 8, 8, 2, 4, 8 };
        uint32_t nFields;
        memcpy(&nFields, all.data() + 14, 4);
        size_t pos = 18;
        for (uint32_t i = 0; i < nFields; i++) {
            uint16_t nameLength, ndim;
            uint8_t dtype;
            uint64_t offset;
            if (pos + 2 > all.size())
                return bad("truncated field table");
            memcpy(&nameLength, all.data() + pos, 2);
            pos += 2;
            if (pos + nameLength + 11 > all.size())
                return bad("truncated field table");
            std::string name(reinterpret_cast<const char*>(all.data() + pos), nameLength);
            pos += nameLength;
            memcpy(&ndim, all.data() + pos, 2);
            pos += 2;
            dtype = all[pos++];
            memcpy(&offset, all.data() + pos, 8);
            pos += 8;
            if (dtype == 0 || dtype > 11)
                return bad("unknown field type");
            TensorField field;
            field.dtype = dtype;
            size_t total = typeSize[dtype];
            for (uint16_t j = 0; j < ndim; j++) {
                uint64_t s;
                if (pos + 8 > all.size())
                    return bad("truncated field table");
                memcpy(&s, all.data() + pos, 8);
                pos += 8;
                field.shape.push_back(size_t(s));
                total *= size_t(s);
            }
            if (offset + total > all.size())
                return bad("field data lies outside the file");
            field.bytes.assign(all.begin() + offset, all.begin() + offset + total);
            fields[name] = field;
        }
        return true;
    }

    const TensorField* field(const std::string& name) const
    {
        auto it = fields.find(name);
        return it == fields.end() ? nullptr : &it->second;
    }
};

/* Marginal2D<Dimension>'s constructor (powitacq_rgb.inl:213-310), writing into the float pool */
inline wpt_rgl_warp buildRglWarp(std::vector<float>& pool, unsigned int sizeX, unsigned int sizeY, const float* data,
        unsigned int dims, const unsigned int* paramRes, const float* const* paramValues, bool normalize, bool buildCdf)
{
    wpt_rgl_warp w;
    memset(&w, 0, sizeof(w));
    w.size_x = sizeX;
    w.size_y = sizeY;
    w.dims = dims;
    w.patch_size[0] = 1.0f / float(sizeX - 1u);
    w.patch_size[1] = 1.0f / float(sizeY - 1u);
    w.inv_patch_size[0] = float(sizeX - 1u);
    w.inv_patch_size[1] = float(sizeY - 1u);
    w.marginal_cdf = w.conditional_cdf = WPT_RGL_NONE;
    uint32_t slices = 1;
    for (int i = int(dims) - 1; i >= 0; --i) {
        w.param_size[i] = paramRes[i];
        w.param_values[i] = uint32_t(pool.size());
        pool.insert(pool.end(), paramValues[i], paramValues[i] + paramRes[i]);
        w.param_stride[i] = paramRes[i] > 1 ? slices : 0;
        slices *= paramRes[i];
    }
    const uint32_t nValues = sizeX * sizeY;
    w.data = uint32_t(pool.size());
    pool.resize(pool.size() + size_t(slices) * nValues);
    if (buildCdf) {
        w.marginal_cdf = uint32_t(pool.size());
        pool.resize(pool.size() + size_t(slices) * sizeY);
        w.conditional_cdf = uint32_t(pool.size());
        pool.resize(pool.size() + size_t(slices) * nValues);
        float* marginal = pool.data() + w.marginal_cdf;
        float* conditional = pool.data() + w.conditional_cdf;
        float* out = pool.data() + w.data;
        for (uint32_t slice = 0; slice < slices; ++slice) {
            for (uint32_t y = 0; y < sizeY; ++y) {
                double sum = 0.0;
                size_t i = size_t(y) * sizeX;
                conditional[i] = 0.0f;
                for (uint32_t x = 0; x < sizeX - 1; ++x, ++i) {
                    sum += 0.5 * (double(data[i]) + double(data[i + 1]));
                    conditional[i + 1] = float(sum);
                }
            }
            marginal[0] = 0.0f;
            double sum = 0.0;
            for (uint32_t y = 0; y < sizeY - 1; ++y) {
                sum += 0.5 * (double(conditional[(y + 1) * sizeX - 1]) + double(conditional[(y + 2) * sizeX - 1]));
                marginal[y + 1] = float(sum);
            }
            const float normalization = 1.0f / marginal[sizeY - 1];
            for (size_t i = 0; i < nValues; ++i)
                conditional[i] *= normalization;
            for (size_t i = 0; i < sizeY; ++i)
                marginal[i] *= normalization;
            for (size_t i = 0; i < nValues; ++i)
                out[i] = data[i] * normalization;
            marginal += sizeY;
            conditional += nValues;
            out += nValues;
            data += nValues;
        }
    } else {
        float* out = pool.data() + w.data;
        for (uint32_t slice = 0; slice < slices; ++slice) {
            float normalization = 1.0f / (w.inv_patch_size[0] * w.inv_patch_size[1]);
            if (normalize) {
                double sum = 0.0;
                for (uint32_t y = 0; y < sizeY - 1; ++y) {
                    size_t i = size_t(y) * sizeX;
                    for (uint32_t x = 0; x < sizeX - 1; ++x, ++i) {
                        float v00 = data[i], v10 = data[i + 1], v01 = data[i + sizeX], v11 = data[i + 1 + sizeX];
                        float avg = 0.25f * (v00 + v10 + v01 + v11);
                        sum += double(avg);
                    }
                }
                normalization = float(1.0 / sum);
            }
            for (uint32_t k = 0; k < nValues; ++k)
                out[k] = data[k] * normalization;
            data += nValues;
            out += nValues;
        }
    }
    return w;
}

/* BRDF::BRDF (powitacq_rgb.inl:893-1003): checks the file structure and builds the five tables.
 * Returns false with a message for files the model rejects. */
inline bool buildRglBrdf(const std::string& filename, std::vector<float>& pool, wpt_rgl_brdf& out, std::string& error)
{
    TensorFile tf;
    if (!tf.load(filename)) {
        error = tf.error;
        return false;
    }
    const TensorField *theta = tf.field("theta_i"), *phi = tf.field("phi_i"), *ndf = tf.field("ndf"), *sigma = tf.field("sigma"),
                      *vndf = tf.field("vndf"), *rgb = tf.field("rgb"), *lum = tf.field("luminance"), *desc = tf.field("description"),
                      *jac = tf.field("jacobian");
    const unsigned int U8 = 1, F32 = 10;
    if (!(theta && phi && ndf && sigma && vndf && rgb && lum && desc && jac
                && desc->shape.size() == 1 && desc->dtype == U8
                && theta->shape.size() == 1 && theta->dtype == F32 && phi->shape.size() == 1 && phi->dtype == F32
                && ndf->shape.size() == 2 && ndf->dtype == F32 && sigma->shape.size() == 2 && sigma->dtype == F32
                && vndf->shape.size() == 4 && vndf->dtype == F32 && vndf->shape[0] == phi->shape[0] && vndf->shape[1] == theta->shape[0]
                && lum->shape.size() == 4 && lum->dtype == F32 && lum->shape[0] == phi->shape[0] && lum->shape[1] == theta->shape[0]
                && lum->shape[2] == lum->shape[3]
                && rgb->dtype == F32 && rgb->shape.size() == 5 && rgb->shape[0] == phi->shape[0] && rgb->shape[1] == theta->shape[0]
                && rgb->shape[2] == 3 && rgb->shape[3] == lum->shape[2] && lum->shape[3] == rgb->shape[4]
                && jac->shape.size() == 1 && jac->shape[0] == 1 && jac->dtype == U8)) {
        error = filename + ": invalid file structure for a measured BRDF";
        return false;
    }
    memset(&out, 0, sizeof(out));
    out.isotropic = phi->shape[0] <= 2 ? 1u : 0u;
    out.jacobian = jac->bytes[0];
    if (!out.isotropic) {
        const float* p = phi->floats();
        const float pi = 3.1415926535897932384626433832795f;
        int reduction = int(std::rint((2 * pi) / (p[phi->shape[0] - 1] - p[0])));
        if (reduction != 1) {
            error = filename + ": reduction != 1 is not supported by the model";
            return false;
        }
    }
    out.ndf = buildRglWarp(pool, ndf->shape[1], ndf->shape[0], ndf->floats(), 0, nullptr, nullptr, false, false);
    out.sigma = buildRglWarp(pool, sigma->shape[1], sigma->shape[0], sigma->floats(), 0, nullptr, nullptr, false, false);
    const unsigned int res2[2] = { (unsigned int)phi->shape[0], (unsigned int)theta->shape[0] };
    const float* val2[2] = { phi->floats(), theta->floats() };
    out.vndf = buildRglWarp(pool, vndf->shape[3], vndf->shape[2], vndf->floats(), 2, res2, val2, true, true);
    out.luminance = buildRglWarp(pool, lum->shape[3], lum->shape[2], lum->floats(), 2, res2, val2, true, true);
    const float channels[3] = { 0.0f, 1.0f, 2.0f };
    const unsigned int res3[3] = { (unsigned int)phi->shape[0], (unsigned int)theta->shape[0], 3u };
    const float* val3[3] = { phi->floats(), theta->floats(), channels };
    out.rgb = buildRglWarp(pool, rgb->shape[4], rgb->shape[3], rgb->floats(), 3, res3, val3, false, false);
    return true;
}

}
