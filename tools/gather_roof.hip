// Microbenchmark: the rate at which a CU serves DEPENDENT per-lane record fetches, the access pattern of the ordered BVH
// traversal (every lane follows its own chain: the next record's index comes out of the record just loaded).
// Build:  hipcc -O3 --offload-arch=gfx950 -o build/gather_roof tools/gather_roof.hip
// Run:    build/gather_roof            (prints one JSON line per case)
//
// A record is 64 bytes (one navigation node).  Variants of how a lane gets it:
//   g4   four 16-byte loads of the lane's own record (what ClosestLane::inner does)
//   g2   two 16-byte loads (a 32-byte record)
//   g1   one 16-byte load (a 16-byte record)
//   quad the four lanes of a quad load the quad's four records together, piece (lane & 3) of record j in instruction j:
//        16 distinct 64-byte lines per instruction instead of 64 (nothing is transposed back: memory pattern only)
//   lds  the table sits in LDS (four ds_read_b128)
//   mix  the first `lds_share` of every 16 steps read LDS, the others memory
//   same g4 with all 64 lanes of a wave on the same record (coherent camera rays near the root)
//   few  g4 with four distinct records per wave (16 lanes each)
//   half / quarter   g4 with every second / fourth lane taking part (the others idle in the same wave)
//   lanemix          g4 where, per step, half of the lanes (chosen per lane and step) read their record from LDS instead
//   rand50 / rand25  g4 with a random half / quarter of the lanes taking part in each step
//   q1   one 16-byte load per lane, the four lanes of a quad on the four pieces of one 64-byte record (16 lines per instruction)
// Bare chains (no arithmetic beside the loads), 128-thread workgroups, `wpe` waves per SIMD resident.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(1);                                                             \
        }                                                                             \
    } while (0)

constexpr int kBlock = 128;
constexpr int kLdsRecs = 128;  // 8 KB table in LDS per workgroup

enum Mode { G4 = 0, G2 = 1, G1 = 2, QUAD = 3, LDS = 4, MIX = 5, SAME = 6, FEW = 7, Q1 = 8, HALF = 9, QUARTER = 10, LANEMIX = 11, RAND50 = 12, RAND25 = 13 };

__device__ __forceinline__ float spin(float x, int valu) {
    for (int i = 0; i < valu; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    return x;
}

// table[i] = 4 x uint4; every piece's .w holds a pseudo-random next index (mask applied by the kernel)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_chain(const uint4* __restrict__ table, uint32_t mask, uint32_t steps, int valu, int lds_share, uint32_t* out) {
    __shared__ uint4 s_tab[kLdsRecs * 4];
    if (MODE == LDS || MODE == MIX || MODE == LANEMIX) {
        for (int i = threadIdx.x; i < kLdsRecs * 4; i += kBlock) s_tab[i] = table[i];
        __syncthreads();
    }
    uint32_t idx = (blockIdx.x * kBlock + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    float f = float(threadIdx.x);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t s = 0; s < steps; ++s) {
        uint32_t nxt;
        if (MODE == RAND50 || MODE == RAND25) {
            nxt = idx * 747796405u + 2891336453u;
            if (((idx >> 9) & (MODE == RAND50 ? 1u : 3u)) == 0u) {
                const uint4* p = table + 4 * size_t(idx & mask);
                const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
                acc += a.x ^ b.y ^ c.z;
                nxt = d.w + a.w + b.w + c.w;
            }
        } else if (MODE == HALF || MODE == QUARTER) {
            nxt = idx;
            if ((lane & (MODE == HALF ? 1u : 3u)) == 0u) {
                const uint4* p = table + 4 * size_t(idx & mask);
                const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
                acc += a.x ^ b.y ^ c.z;
                nxt = d.w + a.w + b.w + c.w;
            }
        } else if (MODE == LANEMIX) {
            if (((idx >> 13) & 1u) != 0u) {
                const uint4* p = s_tab + 4 * (idx & (kLdsRecs - 1));
                const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
                acc += a.x ^ b.y ^ c.z;
                nxt = d.w + a.w + b.w + c.w;
            } else {
                const uint4* p = table + 4 * size_t(idx & mask);
                const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
                acc += a.x ^ b.y ^ c.z;
                nxt = d.w + a.w + b.w + c.w;
            }
        } else if (MODE == G4 || MODE == SAME || MODE == FEW) {
            const uint4* p = table + 4 * size_t(idx & mask);
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x ^ b.y ^ c.z;
            nxt = d.w + a.w + b.w + c.w;
        } else if (MODE == G2) {
            const uint4* p = table + 2 * size_t(idx & mask);
            const uint4 a = p[0], b = p[1];
            acc += a.x ^ b.y;
            nxt = a.w + b.w;
        } else if (MODE == G1) {
            const uint4 a = table[size_t(idx & mask)];
            acc += a.x;
            nxt = a.w;
        } else if (MODE == Q1) {
            const uint32_t qi = uint32_t(__shfl(int(idx), int(lane & ~3u)));
            const uint4 a = table[4 * size_t(qi & mask) + (lane & 3u)];
            acc += a.x;
            nxt = a.w;
        } else if (MODE == QUAD) {
            uint32_t sum = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t other = uint32_t(__shfl(int(idx), int((lane & ~3u) + j)));
                const uint4 a = table[4 * size_t(other & mask) + (lane & 3u)];
                acc += a.x;
                sum += a.w;
            }
            nxt = sum;
        } else {
            const bool from_lds = MODE == LDS || int(s & 15u) < lds_share;
            if (from_lds) {
                const uint4* p = s_tab + 4 * (idx & (kLdsRecs - 1));
                const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
                acc += a.x ^ b.y ^ c.z;
                nxt = d.w + a.w + b.w + c.w;
            } else {
                const uint4* p = table + 4 * size_t(idx & mask);
                const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
                acc += a.x ^ b.y ^ c.z;
                nxt = d.w + a.w + b.w + c.w;
            }
        }
        f = spin(f + float(nxt & 1u), valu);
        idx = nxt * 2654435761u + (MODE == SAME ? 0u : MODE == FEW ? (lane >> 4) * 40503u : lane * 40503u);
        if (MODE == SAME || MODE == FEW) idx = uint32_t(__shfl(int(idx), int(MODE == SAME ? 0u : (lane & 48u))));
    }
    if (acc == 0x12345678u && f == 3.0f) out[0] = acc;  // keeps the chain alive
}

int main(int argc, char** argv) {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const uint32_t steps = argc > 1 ? uint32_t(std::atoi(argv[1])) : 2048u;
    uint32_t* out;
    CHECK(hipMalloc(&out, 64));
    const size_t max_recs = size_t(1) << 24;  // 1 GiB of 64-byte records
    std::vector<uint4> host(max_recs * 4);
    std::mt19937 rng(12345);
    for (auto& v : host) v = make_uint4(rng(), rng(), rng(), rng());
    uint4* table;
    CHECK(hipMalloc(&table, host.size() * sizeof(uint4)));
    CHECK(hipMemcpy(table, host.data(), host.size() * sizeof(uint4), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char* names[] = {"g4", "g2", "g1", "quad", "lds", "mix", "same", "few", "q1", "half", "quarter", "lanemix", "rand50", "rand25"};
    struct Case {
        int mode, log2_recs, valu, wpe, lds_share;
    };
    std::vector<Case> cases;
    for (int lg : {8, 14, 19}) {
        cases.push_back({SAME, lg, 0, 6, 0});
        cases.push_back({FEW, lg, 0, 6, 0});
        cases.push_back({G4, lg, 0, 6, 0});
        cases.push_back({HALF, lg, 0, 6, 0});
        cases.push_back({QUARTER, lg, 0, 6, 0});
        cases.push_back({LANEMIX, lg, 0, 6, 0});
        cases.push_back({RAND50, lg, 0, 6, 0});
        cases.push_back({RAND25, lg, 0, 6, 0});
        cases.push_back({Q1, lg, 0, 6, 0});
        cases.push_back({G1, lg, 0, 6, 0});
    }
    for (int valu : {0})
        for (int wpe : {6, 7}) {
            for (int lg : {8, 14, 19, 24}) {  // 16 KB (L1), 1 MB (L2), 32 MB (all L2s / MALL), 1 GB (HBM)
                cases.push_back({G4, lg, valu, wpe, 0});
                cases.push_back({QUAD, lg, valu, wpe, 0});
                if (wpe == 6) {
                    cases.push_back({G2, lg, valu, wpe, 0});
                    cases.push_back({G1, lg, valu, wpe, 0});
                }
            }
            cases.push_back({LDS, 8, valu, wpe, 16});
            if (wpe == 6)
                for (int share : {4, 8})
                    for (int lg : {14, 19}) cases.push_back({MIX, lg, valu, wpe, share});
        }
    for (const Case& c : cases) {
        const uint32_t mask = (1u << c.log2_recs) - 1u;
        // waves per SIMD -> 128-thread workgroups per CU = 2 * wpe; the grid holds exactly one resident wave set
        const int blocks = cus * 2 * c.wpe;
        auto launch = [&]() {
            switch (c.mode) {
                case G4: k_chain<G4><<<blocks, kBlock, 0, 0>>>(table, mask, steps, c.valu, 0, out); break;
                case G2: k_chain<G2><<<blocks, kBlock, 0, 0>>>(table, mask, steps, c.valu, 0, out); break;
                case G1: k_chain<G1><<<blocks, kBlock, 0, 0>>>(table, mask, steps, c.valu, 0, out); break;
                case QUAD: k_chain<QUAD><<<blocks, kBlock, 0, 0>>>(table, mask, steps, c.valu, 0, out); break;
                case SAME: k_chain<SAME><<<blocks, kBlock, 0, 0>>>(table, mask, steps, c.valu, 0, out); break;
                case Q1: k_chain<Q1><<<blocks, kBlock, 0, 0>>>(table, mask, steps, c.valu, 0, out); break;
                case HALF: k_chain<HALF><<<blocks, kBlock, 0, 0>>>(table, mask, steps, c.valu, 0, out); break;
                case QUARTER: k_chain<QUARTER><<<blocks, kBlock, 0, 0>>>(table, mask, steps, c.valu, 0, out); break;
                case LANEMIX: k_chain<LANEMIX><<<blocks, kBlock, 0, 0>>>(table, mask, steps, c.valu, 0, out); break;
                case RAND50: k_chain<RAND50><<<blocks, kBlock, 0, 0>>>(table, mask, steps, c.valu, 0, out); break;
                case RAND25: k_chain<RAND25><<<blocks, kBlock, 0, 0>>>(table, mask, steps, c.valu, 0, out); break;
                case FEW: k_chain<FEW><<<blocks, kBlock, 0, 0>>>(table, mask, steps, c.valu, 0, out); break;
                case LDS: k_chain<LDS><<<blocks, kBlock, 0, 0>>>(table, mask, steps, c.valu, 16, out); break;
                default: k_chain<MIX><<<blocks, kBlock, 0, 0>>>(table, mask, steps, c.valu, c.lds_share, out); break;
            }
        };
        launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double lane_steps = double(blocks) * kBlock * steps;
        const double per_s = lane_steps / (ms * 1e-3);
        std::printf("{\"mode\": \"%s\", \"table_bytes\": %zu, \"valu_per_step\": %d, \"waves_per_simd\": %d, \"lds_share_of_16\": %d, \"ms\": %.3f, "
                    "\"lane_records_per_s\": %.4g, \"lane_records_per_cu_per_us\": %.1f, \"bytes_per_s\": %.4g}\n",
                    names[c.mode], (size_t(1) << c.log2_recs) * 64, c.valu, c.wpe, (c.mode == LDS || c.mode == MIX) ? c.lds_share : 0, ms, per_s, per_s / cus / 1e6,
                    per_s * (c.mode == G2 ? 32 : (c.mode == G1 || c.mode == Q1) ? 16 : 64));
        std::fflush(stdout);
    }
    return 0;
}
