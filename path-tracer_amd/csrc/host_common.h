// Host-side state shared by the translation units of libpt_hip.so (defined in pt_hip.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <string>

namespace pth {

int fail(int code, const std::string& msg);  // records the message pt_hip_last_error() returns; returns code
hipStream_t stream();                        // the stream pt_hip_init created
int device();                                // the device ordinal pt_hip_init selected, -1 before

}  // namespace pth
