// Host-side state shared by the translation units of libpt_hip.so (defined in pt_hip.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

struct pt_bvh_node;

namespace pth {

int fail(int code, const std::string& msg);  // records the message pt_hip_last_error() returns; returns code
int bind_default();                          // makes pt_hip_init's device and stream the calling thread's current pair
hipStream_t stream();                        // the calling thread's current stream (of the handle or of pt_hip_init)
int device();                                // its device ordinal, -1 when there is none

// pt_hip_build_bvh on the calling thread's current device and stream (pt_hip_scene_create_on builds on the handle's)
int build_bvh_on_current(const float* verts, uint32_t n_verts, const uint32_t* vi, uint32_t n_tris, pt_bvh_node** nodes, uint32_t* n_nodes,
                         uint32_t** prim_ids, uint32_t* max_depth);

}  // namespace pth
