// Internal declarations shared by the host-substrate translation units.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/pt_host.h"

namespace pt_host {

void build_bvh(const float* verts, const uint32_t* vi, uint32_t n_tris, std::vector<pt_bvh_node>& nodes,
               std::vector<uint32_t>& prim_ids, uint32_t& max_depth);

void make_camera(uint32_t width, uint32_t height, float fovy, const float eye[3], const float lookat[3], const float up[3],
                 pt_camera& out);

float triangle_area(const float* verts, const uint32_t* vi, uint32_t tri);

void filter_table(float* out33);

void set_error(const std::string& msg);

// image_io.cpp -- these throw std::runtime_error on malformed files
void save_png(const std::string& path, const float* rgb, uint32_t w, uint32_t h, bool srgb);
bool read_png_rgb8(const std::string& path, std::vector<uint8_t>& rgb, uint32_t& w, uint32_t& h);
bool read_jpeg_rgb8(const std::vector<uint8_t>& file, std::vector<uint8_t>& rgb, uint32_t& w, uint32_t& h);  // jpeg_decode.cpp
bool load_image_linear(const std::string& path, std::vector<float>& rgb, uint32_t& w, uint32_t& h);

}  // namespace pt_host
