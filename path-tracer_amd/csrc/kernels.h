// Wavefront path-tracing kernels (generate+extend, extend, shade, shadow, film) and the batched ray-query kernels.
//
// One render pass keeps `spp_pass` samples of every pixel in flight.  A sample owns a fixed SLOT:
//     slot = ((tile * 16 + group) * spp_pass + s) * 4 + pixel of the group,  tile = 8x8 pixels, group = 2x2 pixels
// (slot_sample / sample_slot below) so that a wave starts as 16 sample numbers of one group (coherent primary rays).
// Radiance is accumulated per slot in `lbuf` in the reference's order (emission, then direct light, bounce by bounce:
// src/integrator.cpp:58-73).
//
// Queues are SHARDED: the slot range is cut into shards of kShard slots, and every shard is an independent compacting
// queue with its own counters.  From bounce to bounce a path stays in its shard (it moves only when the thinned-out paths of
// a pass are gathered into fewer shards, k_compact); a kernel is launched as grid (blocks, shards) and
// block (x, s) works on entries [x * blockDim, ...) of shard s.  One device-wide counter would serialise every wave
// of a launch on a single address (measured: ~90 returning atomics per microsecond per address, which made queue
// allocation the bottleneck of the first version); with a few thousand shards each counter sees at most
// kShard / 64 atomics per launch, and launches need no exact global count -- only the largest shard fill.
//
// Per bounce:  shade (hit list -> next paths + shadow rays) -> shadow (any-hit, adds direct light) + extend (closest
// hit of the next paths, writes the next hit list: misses are dropped here, so shade only sees lanes with work).
#pragma once

#include "shade.h"
#include "traverse.h"
#include "traverse_wave.h"

namespace ptd {

constexpr uint32_t kShard = 4096;  // slots per shard; a multiple of every block size used below

struct PathQ {
    float4* a;        // {o.x, o.y, o.z, d.x}
    float4* b;        // {d.y, d.z, T.x, T.y}
    float4* c;        // {T.z, brdfPdf, bits(px | py << 16), bits(dim | bounce << 10 | frame << 16)}
    uint32_t* slot;   // sample slot
    uint32_t* count;  // [shard] entries in use
};

struct HitQ {
    float4* h0;       // {bits(path index inside the shard), bits(primitive word: index + kPrim* flags), u, v}
    float4* h1;       // {bits(px | py << 16), bits(dim | bounce << 10 | frame << 16), t, 0}  (copied so that shade can
                      //  start its Sobol' and triangle-record reads without waiting for the path state)
    uint32_t* count;  // [shard]
};

struct ShadowQ {
    float4* a;        // {o.x, o.y, o.z, d.x}
    float4* b;        // {d.y, d.z, tmax, bits(slot)}
    float4* c;        // {contribution rgb (throughput already applied), 0}
    uint32_t* count;  // [shard]
};

// Rays the wave-refill kernels hand back (plain-division rays, order-sensitive results): per shard, {queue position,
// bits(distance bound, +inf = run the generic ordered query)}.
struct DeferQ {
    uint2* e;
    uint32_t* count;  // [shard]
};

// Per-shard counters a kernel clears for a LATER kernel of the same stream (none of them is used by the kernel that
// clears them): saves the separate fill launches between the kernels of a bounce.
struct ZeroList {
    uint32_t* p[5];
};
PTD void zero_counters(const ZeroList& z, uint32_t shard) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
            if (z.p[i]) z.p[i][shard] = 0u;
    }
}

// Hand-off of deferred-ray records between the workgroups of a shard, which may sit on different XCDs (separate L2s).
// A device-scope fence would write back and invalidate the whole L2 -- and with it the BVH working set -- at every
// workgroup end (measured: traversal kernels 1.6x slower).  Instead the few words that cross workgroups are written
// and read with device-scope atomic accesses, which go to the coherent level themselves; the writer only waits for
// its own stores to complete before the workgroup signs off.
PTD void store_shared(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PTD uint32_t load_fresh(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// True in every thread of the block that is the last of its shard (gridDim.x blocks per shard) to get here.
// The deferred-ray records (store_shared) must have reached the coherent level before the shard's `done` count can be
// seen by a workgroup on another XCD.  A workgroup-scope release fence does not wait for outstanding vector stores on
// this target, and an agent-scope one would write back the whole L2, so the wave waits for its own memory operations
// explicitly (vmcnt(0), leaving the other counters alone) before the barrier that precedes the sign-off.
PTD bool last_block_of_shard(uint32_t* done, uint32_t shard) {
    __shared__ uint32_t s_last;
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt = 0 (bits 3:0 and 15:14), expcnt and lgkmcnt untouched
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(done + shard, 1u) == gridDim.x - 1u) ? 1u : 0u;
    __syncthreads();
    return s_last != 0u;
}

// Waves per SIMD the wave-refill kernels are built for.
#ifndef WPE
#define WPE 6
#endif
// the camera-ray kernel carries the sampler and camera state on top of the traversal's: at 6 waves per SIMD it spills
// 87 registers, at 5 it is 10 % (1.6 M triangles) to 30 % (3 K triangles) faster; at 4 about the same as at 5
#ifndef WPE_GEN
#define WPE_GEN 5
#endif
// the occlusion query carries no hit record and no distances: 72 registers without spilling, and a seventh wave per SIMD
// is worth 6 % there (the closest-hit kernel spills 36 registers at 7 and loses 3 %)
#ifndef WPE_SHADOW
#define WPE_SHADOW 7
#endif
constexpr uint32_t kMaxWaveBlocks = 8;  // workgroups per shard of the wave-refill kernels: between 2 and 8, see wave_blocks()
constexpr uint32_t kMinWaveBlocks = 2;
constexpr uint32_t kShadeBlocks = 8;  // most workgroups per shard of k_shade (256 threads each), see shade_blocks()

// Shards are spread over grid.y and grid.z (grid.y alone ends at 65535).
PTD uint32_t shard_index() { return blockIdx.z * gridDim.y + blockIdx.y; }

struct PassArgs {
    const uint32_t* frame_tab;  // [frame_local][1024] Sobol' words of the sample numbers of this pass
    uint32_t spp_pass;          // samples per pixel in this pass
    uint32_t tiles_x, tiles_y;
    uint32_t n_slots;
    // image split: the tiles this process renders (slot order) and each tile's position in that list (-1 = not ours);
    // null = all tiles in row-major order
    const uint32_t* tile_list;
    const int32_t* tile_pos;
    int32_t integrator;
    int32_t traversal;
    uint32_t max_bounce;
    // list mode (pt_hip_li): explicit (pixel, frame_local) pairs instead of the tile mapping
    const int32_t* list_xy;
    const uint32_t* list_frame;
    float* list_pixel_sample;
};

// Slot <-> (tile, pixel of the tile, sample number of the pass).  A tile's 8x8 pixels are cut into square groups of
// w x w pixels (w = 1 << PT_SLOT_GROUP_LOG2: 8, 4, 2 or 1), and
//     slot = ((tile * groups + group) * spp_pass + s) * w*w + pixel of the group
// so that 64 consecutive slots -- what a wave starts with -- are 64 / (w*w) consecutive sample numbers of one group.
// w = 8: a wave starts as the 64 pixels of a tile at one sample number; w = 1: as 64 sample numbers of one pixel.  The
// camera rays of a group walk the tree nearly together (merged loads, every lane in the same phase), while the Sobol'
// numbers of ONE sample number at neighbouring pixels are correlated (the shadow rays of a wave pick the same lights):
// measured at 512-sample passes, w = 8 / 4 / 2 / 1: k_generate_wave 46.5 / 43.9 / 40.9 / 39.4 ms on the 1.6 M-triangle
// interior (+0 / +2.0 / +3.5 / +3.5 % rays per second in all), on veach-mis +0 / -0.6 / -0.2 / -1.2 %.
#ifndef PT_SLOT_GROUP_LOG2
#define PT_SLOT_GROUP_LOG2 1
#endif
constexpr uint32_t kGroupLog2 = PT_SLOT_GROUP_LOG2;          // log2 w
constexpr uint32_t kGroupPixLog2 = 2u * kGroupLog2;           // log2 of the pixels of a group
constexpr uint32_t kGroupsLog2 = 6u - kGroupPixLog2;          // log2 of the groups of a tile
static_assert(kGroupLog2 <= 3u, "a group is at most a tile");
// lane = y * 8 + x inside the tile
PTD void slot_sample(const PassArgs& pa, uint32_t slot, uint32_t& tile_pos, uint32_t& lane, uint32_t& frame) {
    const uint32_t p = slot & ((1u << kGroupPixLog2) - 1u), gs = slot >> kGroupPixLog2;
    const uint32_t tg = gs / pa.spp_pass;
    frame = gs - tg * pa.spp_pass;
    const uint32_t g = tg & ((1u << kGroupsLog2) - 1u);
    tile_pos = tg >> kGroupsLog2;
    const uint32_t w_mask = (1u << kGroupLog2) - 1u, row_log2 = 3u - kGroupLog2;  // groups per tile row = 1 << row_log2
    const uint32_t x = ((g & ((1u << row_log2) - 1u)) << kGroupLog2) + (p & w_mask);
    const uint32_t y = ((g >> row_log2) << kGroupLog2) + (p >> kGroupLog2);
    lane = y * 8u + x;
}
PTD uint32_t sample_slot(const PassArgs& pa, uint32_t tile_pos, uint32_t lane, uint32_t frame) {
    const uint32_t x = lane & 7u, y = lane >> 3;
    const uint32_t w_mask = (1u << kGroupLog2) - 1u, row_log2 = 3u - kGroupLog2;
    const uint32_t g = ((y >> kGroupLog2) << row_log2) + (x >> kGroupLog2);
    const uint32_t p = ((y & w_mask) << kGroupLog2) + (x & w_mask);
    return ((((tile_pos << kGroupsLog2) + g) * pa.spp_pass + frame) << kGroupPixLog2) + p;
}

// misc[0] = exact re-traces.  Counting builds only: for kernel class k (0 generate, 1 extend, 2 shadow) the u64 at
// word 4 + 4k holds the node records fetched and the u64 at word 6 + 4k the triangle records fetched.  Words 16..19: the
// queries of k_finish (two u64).  The u64 at word 20 + 2k: those of class k's node records that came from the LDS treelet.
constexpr int kMiscWords = 32;

// A ray that a wave-refill kernel handed back, in reference order: the whole query (plain-division rays, kDeferExact), or
// bounded by `bound` when the ordered pass has already established that the nearest accepted triangle lies before it.
template <int ROWS = kLdsDepth, class C>
PTD bool trace_deferred(const DevScene& sc, const RayD& ray, float bound, HitD& hit, uint32_t* lds, uint32_t* retrace_counter, C& wc) {
    const RayPre pre = make_pre(sc, ray);
    if (bound == kDeferExact || pre.special) return closest_exact<C, ROWS>(sc, ray, hit, lds, wc);
    if (retrace_counter) atomicAdd(retrace_counter, 1u);
    return closest_exact_t<false, C, ROWS>(sc, ray, pre, bound, hit, lds, wc);
}

// add a wave's work counters to the global 64-bit totals (counting builds only)
PTD void flush_counts(const NoCount&, uint32_t*, int) {}
PTD void flush_counts(const Count& c, uint32_t* misc, int kernel_class) {
    if (!misc) return;
    misc += 4 * kernel_class;
    uint32_t n = c.nodes, t = c.tris, l = c.tops;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n += uint32_t(__shfl_xor(int(n), off));
        t += uint32_t(__shfl_xor(int(t), off));
        l += uint32_t(__shfl_xor(int(l), off));
    }
    if (lane_id() == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(misc + 4), (unsigned long long)n);
        atomicAdd(reinterpret_cast<unsigned long long*>(misc + 6), (unsigned long long)t);
        atomicAdd(reinterpret_cast<unsigned long long*>(misc - 4 * kernel_class + 20 + 2 * kernel_class), (unsigned long long)l);
    }
}

// ------------------------------------------------------------------------------------------------ batched queries

// one lane per ray, reference order
template <class C>
__global__ __launch_bounds__(kBlock) void k_intersect(DevScene sc, const float4* rays, uint32_t n, float4* hits, uint32_t* misc) {
    __shared__ uint32_t s_stack[kLdsDepth * kBlock];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    C wc;
    if (i < n) {
        const float4 r0 = rays[2 * size_t(i)], r1 = rays[2 * size_t(i) + 1];
        const RayD ray{V3{r0.x, r0.y, r0.z}, V3{r0.w, r1.x, r1.y}, r1.z, r1.w};
        HitD h;
        closest_exact(sc, ray, h, s_stack + threadIdx.x, wc);
        hits[i] = make_float4(as_f(uint32_t(prim_index(h.prim))), h.t, h.u, h.v);
    }
    flush_counts(wc, misc, 1);
}

template <class C>
__global__ __launch_bounds__(kBlock) void k_occluded(DevScene sc, const float4* rays, uint32_t n, uint8_t* occ, int32_t* first_prim,
                                                      uint32_t* misc) {
    __shared__ uint32_t s_stack[kLdsDepth * kBlock];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    C wc;
    if (i < n) {
        const float4 r0 = rays[2 * size_t(i)], r1 = rays[2 * size_t(i) + 1];
        const RayD ray{V3{r0.x, r0.y, r0.z}, V3{r0.w, r1.x, r1.y}, r1.z, r1.w};
        int32_t fp = -1;
        const bool hit = any_exact(sc, ray, fp, s_stack + threadIdx.x, wc);
        occ[i] = hit ? 1 : 0;
        if (first_prim) first_prim[i] = hit ? prim_index(fp) : -1;
    }
    flush_counts(wc, misc, 2);
}

// ------------------------------------------------------------------------------------------------ generate + first extend

// renderBlock's per-sample prologue (src/main.cpp:63-68): startPixelSample, pixelSample = pixel + samplePixel2D(),
// then the camera ray and the first Scene::rayIntersect of PathIntegrator::Li (src/integrator.cpp:46-53).
// For the two AOV integrators (src/integrator.cpp:20-43) the result is final and goes straight to lbuf.
// Paths that hit something enter path queue `q` and hit list `hq` at the same position of their slot's shard.
// One lane per sample slot, reference order (the ordered mode uses k_generate_wave).
template <class C>
__global__ __launch_bounds__(kBlock) void k_generate(DevScene sc, PassArgs pa, PathQ q, HitQ hq, float4* lbuf, uint32_t* misc) {
    __shared__ uint32_t s_stack[kLdsDepth * kBlock];
    C wc;
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t shard = slot / kShard;  // wave-uniform: kShard is a multiple of the block size
    bool valid = slot < pa.n_slots;
    uint32_t px = 0, py = 0, frame = 0;
    if (valid) {
        if (pa.list_xy) {
            px = uint32_t(pa.list_xy[2 * size_t(slot)]);
            py = uint32_t(pa.list_xy[2 * size_t(slot) + 1]);
            frame = pa.list_frame[slot];
        } else {
            uint32_t lane, tile_pos;
            slot_sample(pa, slot, tile_pos, lane, frame);
            const uint32_t tile = pa.tile_list ? pa.tile_list[tile_pos] : tile_pos;
            px = (tile % pa.tiles_x) * 8u + (lane & 7u);
            py = (tile / pa.tiles_x) * 8u + (lane >> 3);
        }
        valid = px < sc.cam.width && py < sc.cam.height;
    }
    bool alive = false;
    HitD h{-1, 0.0f, 0.0f, 0.0f};
    RayD ray{};
    if (valid) {
        const V2 j = pixel_jitter(sc, pa.frame_tab + size_t(frame) * 1024u, px, py);
        const V2 ps{float(int(px)) + j.x, float(int(py)) + j.y};
        if (pa.list_pixel_sample) {
            pa.list_pixel_sample[2 * size_t(slot)] = ps.x;
            pa.list_pixel_sample[2 * size_t(slot) + 1] = ps.y;
        }
        camera_ray(sc.cam, ps, ray.o, ray.d, ray.tmin, ray.tmax);
        alive = closest_exact(sc, ray, h, s_stack + threadIdx.x, wc);
        V3 out = mk3(0.0f);
        if (alive && pa.integrator != 0) {
            TriRec t;
            t.load(sc, uint32_t(h.prim));
            if (pa.integrator == 1) {
                Surf it;
                complete(t, h.u, h.v, it);
                out = it.n;
            } else {
                out = surface_material(sc, t, h.u, h.v).kd;
            }
            alive = false;
        }
        lbuf[slot] = make_float4(out.x, out.y, out.z, 0.0f);
    } else if (slot < pa.n_slots) {
        lbuf[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    flush_counts(wc, misc, 0);
    const uint32_t local = wave_alloc(alive, q.count + shard);
    if (alive) {
        const size_t dst = size_t(shard) * kShard + local;
        const uint32_t pix = px | (py << 16), meta = 2u | (0u << 10) | (frame << 16);
        q.a[dst] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.d.x);
        q.b[dst] = make_float4(ray.d.y, ray.d.z, 1.0f, 1.0f);
        q.c[dst] = make_float4(1.0f, 0.0f, as_f(pix), as_f(meta));
        q.slot[dst] = slot;
        hq.h0[dst] = make_float4(as_f(local), as_f(uint32_t(h.prim)), h.u, h.v);
        hq.h1[dst] = make_float4(as_f(pix), as_f(meta), h.t, 0.0f);
    }
}

// ------------------------------------------------------------------------------------------------ generate, wave-refill

// Ordered-mode k_generate: the "queue" of shard s is its kShard sample slots, a lane that takes slot j builds the camera
// ray on the spot (closest_wave), and the result goes where k_generate puts it.
struct GenerateIo {
    static constexpr int kPhaseClass = 0;
    PassArgs pa;
    PathQ q;
    HitQ hq;
    DeferQ dq;
    float4* lbuf;
    struct Carry {
        uint32_t pix, frame;
    };
    // a pass smaller than the one the workspace was laid out for leaves the last shards empty
    PTD uint32_t count(uint32_t shard) const { return shard * kShard < pa.n_slots ? min(kShard, pa.n_slots - shard * kShard) : 0u; }
    PTD bool load(const DevScene& sc, uint32_t shard, uint32_t j, RayD& ray, Carry& c) const {
        const uint32_t slot = shard * kShard + j;
        uint32_t px, py, frame;
        if (pa.list_xy) {
            px = uint32_t(pa.list_xy[2 * size_t(slot)]);
            py = uint32_t(pa.list_xy[2 * size_t(slot) + 1]);
            frame = pa.list_frame[slot];
        } else {
            uint32_t lane, tile_pos;
            slot_sample(pa, slot, tile_pos, lane, frame);
            const uint32_t tile = pa.tile_list ? pa.tile_list[tile_pos] : tile_pos;
            px = (tile % pa.tiles_x) * 8u + (lane & 7u);
            py = (tile / pa.tiles_x) * 8u + (lane >> 3);
        }
        c.pix = px | (py << 16);
        c.frame = frame;
        ray = RayD{V3{0.0f, 0.0f, 0.0f}, V3{1.0f, 1.0f, 1.0f}, 0.0f, 0.0f};
        if (!(px < sc.cam.width && py < sc.cam.height)) return false;
        const V2 jit = pixel_jitter(sc, pa.frame_tab + size_t(frame) * 1024u, px, py);
        const V2 ps{float(int(px)) + jit.x, float(int(py)) + jit.y};
        if (pa.list_pixel_sample) {
            pa.list_pixel_sample[2 * size_t(slot)] = ps.x;
            pa.list_pixel_sample[2 * size_t(slot) + 1] = ps.y;
        }
        camera_ray(sc.cam, ps, ray.o, ray.d, ray.tmin, ray.tmax);
        return true;
    }
    // what k_generate does with the result of slot j: AOVs are final, path-traced samples enter the queues
    PTD void emit(const DevScene& sc, uint32_t shard, bool have, uint32_t j, const HitD& h, const Carry& c, const V3& o, const V3& d) const {
        const size_t sbase = size_t(shard) * kShard;
        bool alive = have && h.prim >= 0;
        if (have) {
            V3 out = mk3(0.0f);
            if (alive && pa.integrator != 0) {
                TriRec t;
                t.load(sc, uint32_t(h.prim));
                if (pa.integrator == 1) {
                    Surf it;
                    complete(t, h.u, h.v, it);
                    out = it.n;
                } else {
                    out = surface_material(sc, t, h.u, h.v).kd;
                }
                alive = false;
            }
            lbuf[sbase + j] = make_float4(out.x, out.y, out.z, 0.0f);
        }
        const uint32_t local = wave_alloc(alive, q.count + shard);
        if (alive) {
            const size_t dst = sbase + local;
            const uint32_t meta = 2u | (0u << 10) | (c.frame << 16);
            q.a[dst] = make_float4(o.x, o.y, o.z, d.x);
            q.b[dst] = make_float4(d.y, d.z, 1.0f, 1.0f);
            q.c[dst] = make_float4(1.0f, 0.0f, as_f(c.pix), as_f(meta));
            q.slot[dst] = uint32_t(sbase + j);
            hq.h0[dst] = make_float4(as_f(local), as_f(uint32_t(h.prim)), h.u, h.v);
            hq.h1[dst] = make_float4(as_f(c.pix), as_f(meta), h.t, 0.0f);
        }
    }
    PTD void commit(const DevScene& sc, uint32_t shard, bool fin, uint32_t j, const HitD& h, bool defer, float bound, const Carry& c, const V3& o,
                    const V3& d) const {
        emit(sc, shard, fin && !defer, j, h, c, o, d);
        const bool df = fin && defer;
        if (wave_any(df)) {
            const uint32_t dl = wave_alloc(df, dq.count + shard);
            if (df) {
                store_shared(&dq.e[size_t(shard) * kShard + dl].x, j);
                store_shared(&dq.e[size_t(shard) * kShard + dl].y, as_u(bound));
            }
        }
    }
};

template <class C, class Lane>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(WPE_GEN, WPE_GEN))) void k_generate_wave(DevScene sc, GenerateIo io,
                                                                                                                  uint32_t* fetch, uint32_t* done,
                                                                                                                  uint32_t* misc) {
    __shared__ uint32_t s_stack[wave_lds_words<Lane>()];
    const uint32_t shard = shard_index();
    C wc;
    __shared__ uint4 s_nodes[treelet_lds_vec4<Lane>()];
    const Treelet tl = wave_treelet<Lane>(sc, s_nodes);
    closest_wave<Lane>(sc, io, shard, fetch, s_stack + threadIdx.x, tl, wc);
    if (last_block_of_shard(done, shard)) {
        const size_t sbase = size_t(shard) * kShard;
        const uint32_t n = load_fresh(io.dq.count + shard);
        for (uint32_t base = 0; base < n; base += kBlock) {
            const uint32_t k = base + threadIdx.x;
            HitD h{-1, 0.0f, 0.0f, 0.0f};
            uint32_t j = 0;
            GenerateIo::Carry c{};
            RayD ray{};
            if (k < n) {
                j = load_fresh(&io.dq.e[sbase + k].x);
                const float bound = as_f(load_fresh(&io.dq.e[sbase + k].y));
                io.load(sc, shard, j, ray, c);
                trace_deferred<Lane::kLdsRows>(sc, ray, bound, h, s_stack + threadIdx.x, misc, wc);
            }
            io.emit(sc, shard, k < n, j, h, c, ray.o, ray.d);
        }
    }
    flush_counts(wc, misc, 0);
}

// ------------------------------------------------------------------------------------------------ wave-refill variants

// Scene::rayIntersect for rays spawned by Intersection::genRay (src/shape.cpp:186-189): tmin 0, tmax +inf.  Rays are taken
// from the shard's queue through a fetch counter (closest_wave, in either traversal order).  Hits are appended to the
// shard's hit list; a miss ends the path (src/integrator.cpp:53) and is simply not listed; deferred rays go to `dq` and are
// finished by the shard's last workgroup.
struct ExtendIo {
    static constexpr int kPhaseClass = 1;
    PathQ q;
    HitQ hq;
    DeferQ dq;
    struct Carry {
        float pix, meta;
    };
    PTD uint32_t count(uint32_t shard) const { return q.count[shard]; }
    PTD bool load(const DevScene&, uint32_t shard, uint32_t j, RayD& ray, Carry& c) const {
        const size_t i = size_t(shard) * kShard + j;
        const float4 a = q.a[i], b = q.b[i], cc = q.c[i];
        ray = RayD{V3{a.x, a.y, a.z}, V3{a.w, b.x, b.y}, 0.0f, __builtin_huge_valf()};
        c.pix = cc.z;
        c.meta = cc.w;
        return true;
    }
    PTD void commit(const DevScene&, uint32_t shard, bool fin, uint32_t j, const HitD& h, bool defer, float bound, const Carry& c, const V3&, const V3&) const {
        const size_t sbase = size_t(shard) * kShard;
        const bool found = fin && !defer && h.prim >= 0;
        const uint32_t local = wave_alloc(found, hq.count + shard);
        if (found) {
            hq.h0[sbase + local] = make_float4(as_f(j), as_f(uint32_t(h.prim)), h.u, h.v);
            hq.h1[sbase + local] = make_float4(c.pix, c.meta, h.t, 0.0f);
        }
        const bool d = fin && defer;
        if (wave_any(d)) {
            const uint32_t dl = wave_alloc(d, dq.count + shard);
            if (d) {
                store_shared(&dq.e[sbase + dl].x, j);
                store_shared(&dq.e[sbase + dl].y, as_u(bound));
            }
        }
    }
};

// The deferred rays of one shard, by one workgroup, in reference order (order-sensitive ones only up to their bound).
template <int ROWS, class C>
PTD void extend_deferred(const DevScene& sc, const PathQ& q, const HitQ& hq, const DeferQ& dq, uint32_t shard, uint32_t* lds_stack,
                         uint32_t* misc, C& wc) {
    const size_t sbase = size_t(shard) * kShard;
    const uint32_t n = load_fresh(dq.count + shard);
    for (uint32_t base = 0; base < n; base += kBlock) {
        const uint32_t k = base + threadIdx.x;
        bool found = false;
        HitD h{-1, 0.0f, 0.0f, 0.0f};
        uint32_t j = 0;
        float4 c{};
        if (k < n) {
            j = load_fresh(&dq.e[sbase + k].x);
            const float bound = as_f(load_fresh(&dq.e[sbase + k].y));
            const float4 a = q.a[sbase + j], b = q.b[sbase + j];
            c = q.c[sbase + j];
            const RayD ray{V3{a.x, a.y, a.z}, V3{a.w, b.x, b.y}, 0.0f, __builtin_huge_valf()};
            found = trace_deferred<ROWS>(sc, ray, bound, h, lds_stack, misc, wc);
        }
        const uint32_t local = wave_alloc(found, hq.count + shard);
        if (found) {
            hq.h0[sbase + local] = make_float4(as_f(j), as_f(uint32_t(h.prim)), h.u, h.v);
            hq.h1[sbase + local] = make_float4(c.z, c.w, h.t, 0.0f);
        }
    }
}

// gridDim.x workgroups share a shard's queue; the last one to run dry also finishes the shard's deferred rays, so that
// this (slow, nearly serial) work overlaps with the other shards instead of forming a tail of its own.
template <class C, class Lane>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_extend_wave(DevScene sc, ExtendIo io, uint32_t* fetch,
                                                                                                    uint32_t* done, ZeroList zl, uint32_t* misc) {
    __shared__ uint32_t s_stack[wave_lds_words<Lane>()];
    const uint32_t shard = shard_index();
    zero_counters(zl, shard);
    C wc;
    __shared__ uint4 s_nodes[treelet_lds_vec4<Lane>()];
    const Treelet tl = wave_treelet<Lane>(sc, s_nodes);
    closest_wave<Lane>(sc, io, shard, fetch, s_stack + threadIdx.x, tl, wc);
    if (last_block_of_shard(done, shard)) extend_deferred<Lane::kLdsRows>(sc, io.q, io.hq, io.dq, shard, s_stack + threadIdx.x, misc, wc);
    flush_counts(wc, misc, 1);
}

// Visibility test of sampleLd (src/integrator.cpp:124-126) through any_wave, in either traversal order: unoccluded rays
// add their contribution to the sample's radiance slot; plain-division rays go to `dq`.
struct ShadowIo {
    static constexpr int kPhaseClass = 2;
    ShadowQ sh;
    DeferQ dq;
    float4* lbuf;
    struct Carry {
        uint32_t slot;
    };
    PTD uint32_t count(uint32_t shard) const { return sh.count[shard]; }
    PTD void load(const DevScene&, uint32_t shard, uint32_t j, RayD& ray, Carry& c) const {
        const size_t i = size_t(shard) * kShard + j;
        const float4 a = sh.a[i], b = sh.b[i];
        ray = RayD{V3{a.x, a.y, a.z}, V3{a.w, b.x, b.y}, 0.0f, b.z};
        c.slot = as_u(b.w);
    }
    PTD void commit(const DevScene&, uint32_t shard, bool fin, uint32_t j, bool occluded, bool defer, const Carry& c) const {
        const size_t sbase = size_t(shard) * kShard;
        if (fin && !defer && !occluded) {
            const float4 v = sh.c[sbase + j];
            float4 L = lbuf[c.slot];
            L.x += v.x;
            L.y += v.y;
            L.z += v.z;
            lbuf[c.slot] = L;
        }
        const bool d = fin && defer;
        if (wave_any(d)) {
            const uint32_t dl = wave_alloc(d, dq.count + shard);
            if (d) store_shared(&dq.e[sbase + dl].x, j);
        }
    }
};

template <class C, class Lane>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(WPE_SHADOW, WPE_SHADOW))) void k_shadow_wave(DevScene sc, ShadowIo io, uint32_t* fetch,
                                                                                                    uint32_t* done, uint32_t* misc) {
    __shared__ uint32_t s_stack[wave_lds_words<Lane>()];
    const uint32_t shard = shard_index();
    C wc;
    __shared__ uint4 s_nodes[treelet_lds_vec4<Lane>()];
    const Treelet tl = wave_treelet<Lane>(sc, s_nodes);
    any_wave<Lane>(sc, io, shard, fetch, s_stack + threadIdx.x, tl, wc);
    if (last_block_of_shard(done, shard)) {  // plain-division rays of the shard
        const size_t sbase = size_t(shard) * kShard;
        const uint32_t n = load_fresh(io.dq.count + shard);
        for (uint32_t k = threadIdx.x; k < n; k += kBlock) {
            const uint32_t j = load_fresh(&io.dq.e[sbase + k].x);
            const float4 a = io.sh.a[sbase + j], b = io.sh.b[sbase + j];
            const RayD ray{V3{a.x, a.y, a.z}, V3{a.w, b.x, b.y}, 0.0f, b.z};
            int32_t first_prim;
            if (!any_exact<C, Lane::kLdsRows>(sc, ray, first_prim, s_stack + threadIdx.x, wc)) {
                const float4 v = io.sh.c[sbase + j];
                const uint32_t slot = as_u(b.w);
                float4 L = io.lbuf[slot];
                L.x += v.x;
                L.y += v.y;
                L.z += v.z;
                io.lbuf[slot] = L;
            }
        }
    }
    flush_counts(wc, misc, 2);
}

// Ordered-mode batched queries: the batch is cut into shards of kShard rays, each with its own fetch counter.  Deferred
// rays are marked in the output (prim = kBatchDeferred, t = bound; occluded = 2) and finished by the *_fix kernels.
constexpr uint32_t kBatchDeferred = 0xfffffffeu;

struct BatchClosestIo {
    static constexpr int kPhaseClass = 3;
    const float4* rays;
    uint32_t n;
    float4* hits;
    struct Carry {};
    PTD uint32_t count(uint32_t shard) const { return min(kShard, n - shard * kShard); }
    PTD bool load(const DevScene&, uint32_t shard, uint32_t j, RayD& ray, Carry&) const {
        const size_t i = size_t(shard) * kShard + j;
        const float4 r0 = rays[2 * i], r1 = rays[2 * i + 1];
        ray = RayD{V3{r0.x, r0.y, r0.z}, V3{r0.w, r1.x, r1.y}, r1.z, r1.w};
        return true;
    }
    PTD void commit(const DevScene&, uint32_t shard, bool fin, uint32_t j, const HitD& h, bool defer, float bound, const Carry&, const V3&, const V3&) const {
        if (!fin) return;
        const size_t i = size_t(shard) * kShard + j;
        hits[i] = defer ? make_float4(as_f(kBatchDeferred), bound, 0.0f, 0.0f) : make_float4(as_f(uint32_t(prim_index(h.prim))), h.t, h.u, h.v);
    }
};

template <class C>
__global__ __launch_bounds__(kBlock) void k_intersect_wave(DevScene sc, BatchClosestIo io, uint32_t* fetch, uint32_t* misc) {
    __shared__ uint32_t s_stack[wave_lds_words<ClosestLane>()];
    C wc;
    __shared__ uint4 s_nodes[treelet_lds_vec4<ClosestLane>()];
    const Treelet tl = wave_treelet<ClosestLane>(sc, s_nodes);
    closest_wave<ClosestLane>(sc, io, blockIdx.y, fetch, s_stack + threadIdx.x, tl, wc);
    flush_counts(wc, misc, 1);
}

template <class C>
__global__ __launch_bounds__(kBlock) void k_intersect_fix(DevScene sc, const float4* rays, uint32_t n, float4* hits, uint32_t* misc) {
    __shared__ uint32_t s_stack[kLdsDepth * kBlock];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    C wc;
    if (i < n) {
        const float4 m = hits[i];
        if (as_u(m.x) == kBatchDeferred) {
            const float4 r0 = rays[2 * size_t(i)], r1 = rays[2 * size_t(i) + 1];
            const RayD ray{V3{r0.x, r0.y, r0.z}, V3{r0.w, r1.x, r1.y}, r1.z, r1.w};
            HitD h;
            trace_deferred(sc, ray, m.y, h, s_stack + threadIdx.x, misc, wc);
            hits[i] = make_float4(as_f(uint32_t(prim_index(h.prim))), h.t, h.u, h.v);
        }
    }
    flush_counts(wc, misc, 1);
}

struct BatchAnyIo {
    static constexpr int kPhaseClass = 3;
    const float4* rays;
    uint32_t n;
    uint8_t* occ;
    struct Carry {};
    PTD uint32_t count(uint32_t shard) const { return min(kShard, n - shard * kShard); }
    PTD void load(const DevScene&, uint32_t shard, uint32_t j, RayD& ray, Carry&) const {
        const size_t i = size_t(shard) * kShard + j;
        const float4 r0 = rays[2 * i], r1 = rays[2 * i + 1];
        ray = RayD{V3{r0.x, r0.y, r0.z}, V3{r0.w, r1.x, r1.y}, r1.z, r1.w};
    }
    PTD void commit(const DevScene&, uint32_t shard, bool fin, uint32_t j, bool occluded, bool defer, const Carry&) const {
        if (fin) occ[size_t(shard) * kShard + j] = defer ? uint8_t(2) : uint8_t(occluded ? 1 : 0);
    }
};

template <class C>
__global__ __launch_bounds__(kBlock) void k_occluded_wave(DevScene sc, BatchAnyIo io, uint32_t* fetch, uint32_t* misc) {
    __shared__ uint32_t s_stack[wave_lds_words<AnyLane>()];
    C wc;
    __shared__ uint4 s_nodes[treelet_lds_vec4<AnyLane>()];
    const Treelet tl = wave_treelet<AnyLane>(sc, s_nodes);
    any_wave<AnyLane>(sc, io, blockIdx.y, fetch, s_stack + threadIdx.x, tl, wc);
    flush_counts(wc, misc, 2);
}

// finishes deferred rays and fills first_prim (ordered mode has no "first found" triangle: -2 for occluded rays)
template <class C>
__global__ __launch_bounds__(kBlock) void k_occluded_fix(DevScene sc, const float4* rays, uint32_t n, uint8_t* occ, int32_t* first_prim,
                                                          uint32_t* misc) {
    __shared__ uint32_t s_stack[kLdsDepth * kBlock];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    C wc;
    if (i < n) {
        uint8_t o = occ[i];
        if (o == 2) {
            const float4 r0 = rays[2 * size_t(i)], r1 = rays[2 * size_t(i) + 1];
            const RayD ray{V3{r0.x, r0.y, r0.z}, V3{r0.w, r1.x, r1.y}, r1.z, r1.w};
            int32_t fp;
            o = any_exact(sc, ray, fp, s_stack + threadIdx.x, wc) ? 1 : 0;
            occ[i] = o;
        }
        if (first_prim) first_prim[i] = o ? -2 : -1;
    }
    flush_counts(wc, misc, 2);
}

// ------------------------------------------------------------------------------------------------ shade

// What one pass of the loop body hands on: the continued path (next ray, throughput, sampler position) and the shadow ray
// with its already-weighted contribution.
struct ShadeOut {
    bool cont = false, want_shadow = false;
    float4 na{}, nb{}, nc{}, sa{}, sb{}, sc4{};
};

// One iteration of the loop body of PathIntegrator::Li after the intersection (src/integrator.cpp:55-103) plus
// sampleLd up to the visibility test (src/integrator.cpp:109-141) for one hit: h0 / h1 are its hit-list entry, a / b / c
// the state of the path that found it, slot its radiance slot (emitted radiance is added there right away).
//
// Memory reads are arranged in dependent levels: (1) the hit-list entry; (2) the path state, the triangle's shading
// record and the Sobol' window (all addressed from the entry); (3) the record of the light chosen by the first Sobol'
// value, read by light_sample_at.
PTD void shade_hit(const DevScene& sc, const PassArgs& pa, const float4 h0, const float4 h1, const float4 a, const float4 b, const float4 c,
                   const uint32_t slot, float4* lbuf, ShadeOut& o) {
    const uint32_t prim = as_u(h0.y);
    const uint32_t pix = as_u(h1.x), meta = as_u(h1.y);
    const uint32_t frame = meta >> 16;
    int bounce = int((meta >> 10) & 63u);
    TriRec tr;
    tr.load(sc, prim);
    Sampler smp;
    smp.start(sc, pa.frame_tab + size_t(frame) * 1024u, pix & 0xffffu, pix >> 16, int(meta & 1023u));
    // level 3: the light record depends on the first Sobol' value only
    const float4* lrec_ptr = nullptr;
    float u1 = 0.0f;
    if (sc.n_lights != 0u) {
        u1 = smp.s1();
        uint32_t li = uint32_t(int(u1 * float(sc.n_lights)));  // include/pt/light.h:58-60
        if (li >= sc.n_lights) li = sc.n_lights - 1u;          // the reference would throw here; clamp instead
        lrec_ptr = sc.lrec + size_t(li) * kLightRecVec4;
    }

    const V3 ro{a.x, a.y, a.z}, rd{a.w, b.x, b.y};
    V3 T{b.z, b.w, c.x};
    const float brdfPdf = c.y;
    Surf it;
    complete(tr, h0.z, h0.w, it);
    const V3 wo = -rd;
    const Mat m = surface_material(sc, tr, h0.z, h0.w);

    // emitted radiance with MIS against light sampling (src/integrator.cpp:58-69)
    if (tr.light_id() >= 0) {
        const V3 Le = dot(it.n, wo) > 0.0f ? tr.lemit() : mk3(0.0f);
        V3 add;
        if (bounce == 0) {
            add = cmul(T, Le);
        } else {
            float lpdf = light_pdf(tr.inv_area(), it.p, it.n, ro, rd);
            lpdf *= 1.0f / float(sc.n_lights);
            const float w = power_heuristic(brdfPdf, lpdf);
            add = w * cmul(T, Le);
        }
        float4 L = lbuf[slot];
        L.x += add.x;
        L.y += add.y;
        L.z += add.z;
        lbuf[slot] = L;
    }

    // light sampling (src/integrator.cpp:109-141)
    if (sc.n_lights != 0u) {
        const float selectPdf = 1.0f / float(sc.n_lights);
        const LiSample ls = light_sample_at(lrec_ptr, it.p, it.n, smp.s2());
        if (ls.pdfDir != 0.0f) {
            // Scene::unocculded (src/scene.cpp:213-220)
            const V3 p0 = it.p + it.n * kEps, p1 = ls.p + ls.n * kEps;
            const V3 d = p1 - p0;
            const float dist = norm(d);
            const V3 sd = d / dist;
            V3 f;
            float bp;
            brdf_eval_pdf(m, it.n, wo, ls.wi, f, bp);
            const float cosTheta = dot(it.n, ls.wi);
            const float lp = ls.pdfDir * selectPdf;
            const float w = power_heuristic(lp, bp);
            const V3 Ld = w * cmul(f, ls.L) * cosTheta / lp;
            const V3 contrib = cmul(T, Ld);
            o.want_shadow = true;
            o.sa = make_float4(p0.x, p0.y, p0.z, sd.x);
            o.sb = make_float4(sd.y, sd.z, dist * (1 - kEps), as_f(slot));
            o.sc4 = make_float4(contrib.x, contrib.y, contrib.z, 0.0f);
        }
    }

    // BRDF sampling; the 2-D sample is drawn before the 1-D one (src/integrator.cpp:83, right-to-left arguments)
    const V2 u = smp.s2();
    const float uc = smp.s1();
    const BSample bs = brdf_sample(m, it, wo, uc, u);
    float newPdf = brdfPdf;
    bool go = true;
    if (bs.specular) {
        bounce--;
        newPdf = 1.0f;
    } else {
        if (sqnorm(bs.f) == 0.0f || bs.pdf == 0.0f) go = false;
        if (go) {
            const V3 thr = bs.f * dot(it.n, bs.wi) / bs.pdf;
            T = cmul(T, thr);
            newPdf = bs.pdf;
        }
    }
    if (go) {
        const V3 no = it.p + it.ng * kEps;  // Intersection::genRay (src/shape.cpp:186-189)
        // Russian roulette (src/integrator.cpp:99-103)
        if (maxc(T) < 1.0f && bounce > 1) {
            const float qq = std_max(0.0f, 1.0f - maxc(T));
            if (smp.s1() < qq) go = false;
            T = T / (1.0f - qq);
        }
        bounce++;
        if (go && bounce < int(pa.max_bounce)) {
            o.cont = true;
            o.na = make_float4(no.x, no.y, no.z, bs.wi.x);
            o.nb = make_float4(bs.wi.y, bs.wi.z, T.x, T.y);
            // the sampler wraps lazily (dim == 1024 after drawing dimension 1023): store the dimension the next
            // draw would wrap to (src/sampler.cpp:102-114), so that it fits the 10-bit field
            const uint32_t next_dim = smp.dim >= 1024 ? 2u : uint32_t(smp.dim);
            o.nc = make_float4(T.z, newPdf, as_f(pix), as_f(next_dim | (uint32_t(bounce) << 10) | (frame << 16)));
        }
    }
}

// shade_hit for every entry of the hit lists: the shadow ray goes to the shadow queue, the continued path to the next
// path queue.  120 registers, 4 waves per SIMD, thanks to three real calls (pow_k, sincos_call, light_sample_at); capping
// the registers of the fully inlined code instead spills and is much slower.
__global__ __launch_bounds__(256) void k_shade(DevScene sc, PassArgs pa, HitQ hq, PathQ in, PathQ out, ShadowQ sh, float4* lbuf, ZeroList zl) {
    const uint32_t shard = shard_index();
    zero_counters(zl, shard);
    const size_t sbase = size_t(shard) * kShard;
    const uint32_t n = hq.count[shard];
    // the hit-list entry of the next iteration is requested before the current one is shaded: one dependent level less
    float4 h0n{}, h1n{};
    if (blockIdx.x * 256u + threadIdx.x < n) {
        h0n = hq.h0[sbase + blockIdx.x * 256u + threadIdx.x];
        h1n = hq.h1[sbase + blockIdx.x * 256u + threadIdx.x];
    }
    for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {
        const uint32_t j = base + threadIdx.x;
        ShadeOut o;
        uint32_t slot = 0;
        const float4 h0 = h0n, h1 = h1n;
        const uint32_t jn = j + gridDim.x * 256u;
        if (jn < n) {
            h0n = hq.h0[sbase + jn];
            h1n = hq.h1[sbase + jn];
        }
        if (j < n) {
            const size_t pi = sbase + as_u(h0.x);
            const float4 a = in.a[pi], b = in.b[pi], c = in.c[pi];  // level 2 reads
            slot = in.slot[pi];
            shade_hit(sc, pa, h0, h1, a, b, c, slot, lbuf, o);
        }
        const uint32_t sl = wave_alloc(o.want_shadow, sh.count + shard);
        if (o.want_shadow) {
            sh.a[sbase + sl] = o.sa;
            sh.b[sbase + sl] = o.sb;
            sh.c[sbase + sl] = o.sc4;
        }
        const uint32_t nl = wave_alloc(o.cont, out.count + shard);
        if (o.cont) {
            out.a[sbase + nl] = o.na;
            out.b[sbase + nl] = o.nb;
            out.c[sbase + nl] = o.nc;
            out.slot[sbase + nl] = slot;
        }
    }
}

// ------------------------------------------------------------------------------------------------ finish

// Up to `budget` further iterations of a few paths in one launch.  Two situations call for it.  (1) The tail of every
// pass: Russian roulette thins the paths out quickly (after ten bounces one in two thousand is left), but every further
// bounce still costs three launches whose time no longer depends on how little they carry -- a tenth of a pass went into
// its last twenty bounces.  (2) A specular bounce is not counted (src/integrator.cpp:84-87), so once every path has had
// maxBounce iterations only chains of mirror reflections are left, often a handful of paths that go on for thousands of
// iterations (a mirror with interpolated normals reflects rays below its own surface, and such a path creeps along a
// crease).
// Here the first `lanes` lanes of a wave take one path each -- entry `j` of the hit list of shard `shard`, both from
// `entries` -- and run shade, visibility test and next intersection, all in the reference's visit order, until the path
// ends: a chain of dependent memory accesses either way.  Paths in the lanes of one wave are in different phases all the
// time, so few lanes per wave keep the iteration short (measured: 10 us per iteration with one path per wave, 18 us
// with 64): one for the long chains, four for the tail of a pass, where the number of paths the chip holds at once
// counts for more.  A path that is still alive when the
// budget is used up goes back to its shard: its state to path queue `out`, its pending hit to hit list `hq_out`, as the
// shade and extend kernels would have left them.  misc[16..19] receive the numbers of closest-hit and any-hit queries
// as two 64-bit totals (without the pending hits that go back: those are counted with the queue they go to).
constexpr int kFinishBlock = 64;
template <class C>
__global__ __launch_bounds__(kFinishBlock) void k_finish(DevScene sc, PassArgs pa, HitQ hq, PathQ q, PathQ out, HitQ hq_out, const uint2* entries,
                                                         uint32_t n_entries, uint32_t lanes, float4* lbuf, uint32_t budget, uint32_t* misc) {
    __shared__ uint32_t s_stack[kLdsDepth * kBlock];  // the stack's rows are kBlock words apart
    const uint32_t ei = blockIdx.x * lanes + threadIdx.x;
    const bool mine = threadIdx.x < lanes && ei < n_entries;
    const uint2 e = mine ? entries[ei] : make_uint2(0u, 0xffffffffu);
    const size_t sbase = size_t(e.x) * kShard;
    C wce, wcs;
    uint32_t n_closest = 0, n_any = 0;
    uint32_t* const my_stack = s_stack + threadIdx.x;
    if (mine && e.y < hq.count[e.x]) {
        float4 h0 = hq.h0[sbase + e.y], h1 = hq.h1[sbase + e.y];
        const size_t pi = sbase + as_u(h0.x);
        float4 a = q.a[pi], b = q.b[pi], c = q.c[pi];
        const uint32_t slot = q.slot[pi];
        bool alive = true;
        for (uint32_t it = 0; it < budget; ++it) {
            ShadeOut o;
            shade_hit(sc, pa, h0, h1, a, b, c, slot, lbuf, o);
            if (o.want_shadow) {
                const RayD sr{V3{o.sa.x, o.sa.y, o.sa.z}, V3{o.sa.w, o.sb.x, o.sb.y}, 0.0f, o.sb.z};
                int32_t first_prim;
                ++n_any;
                if (!any_exact(sc, sr, first_prim, my_stack, wcs)) {
                    float4 L = lbuf[slot];
                    L.x += o.sc4.x;
                    L.y += o.sc4.y;
                    L.z += o.sc4.z;
                    lbuf[slot] = L;
                }
            }
            alive = false;
            if (!o.cont) break;
            const RayD ray{V3{o.na.x, o.na.y, o.na.z}, V3{o.na.w, o.nb.x, o.nb.y}, 0.0f, __builtin_huge_valf()};
            HitD h;
            ++n_closest;
            if (!closest_exact(sc, ray, h, my_stack, wce)) break;
            alive = true;
            a = o.na;
            b = o.nb;
            c = o.nc;
            h0 = make_float4(0.0f, as_f(uint32_t(h.prim)), h.u, h.v);
            h1 = make_float4(c.z, c.w, h.t, 0.0f);
        }
        if (alive) {
            const uint32_t nl = atomicAdd(out.count + e.x, 1u);
            out.a[sbase + nl] = a;
            out.b[sbase + nl] = b;
            out.c[sbase + nl] = c;
            out.slot[sbase + nl] = slot;
            const uint32_t hl = atomicAdd(hq_out.count + e.x, 1u);
            hq_out.h0[sbase + hl] = make_float4(as_f(nl), h0.y, h0.z, h0.w);
            hq_out.h1[sbase + hl] = h1;
            --n_closest;
        }
        atomicAdd(reinterpret_cast<unsigned long long*>(misc + 16), (unsigned long long)n_closest);
        atomicAdd(reinterpret_cast<unsigned long long*>(misc + 18), (unsigned long long)n_any);
    }
    flush_counts(wce, misc, 1);
    flush_counts(wcs, misc, 2);
}

// ------------------------------------------------------------------------------------------------ compact

// Whenever Russian roulette has left a small share of the paths in play: the survivors, a few per shard, are gathered
// into the first shards of the other path queue and of a spare hit list, densely, so that the remaining bounces run on
// a few full shards instead of all of them nearly empty (a launch over 115 000 shards with three rays each costs what
// it costs to start 115 000 workgroups).  Which shard a path sits in has no bearing on its result: its radiance slot
// travels with it.  New positions: an exclusive prefix sum over the shards' hit counts (k_shard_offsets: one workgroup,
// every thread a contiguous run of shards), so the order of the paths is kept and no counter is contended.
constexpr int kOffsetsBlock = 1024;
__global__ __launch_bounds__(kOffsetsBlock) void k_shard_offsets(const uint32_t* counts, uint32_t n_shards, uint32_t* offsets) {
    __shared__ uint32_t s_sum[kOffsetsBlock];
    const uint32_t per = (n_shards + kOffsetsBlock - 1) / kOffsetsBlock;
    const uint32_t lo = min(threadIdx.x * per, n_shards), hi = min(lo + per, n_shards);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += counts[i];
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < kOffsetsBlock; off <<= 1) {  // inclusive scan of the per-thread sums
        const uint32_t v = threadIdx.x >= uint32_t(off) ? s_sum[threadIdx.x - off] : 0u;
        __syncthreads();
        s_sum[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = s_sum[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; ++i) {
        offsets[i] = run;
        run += counts[i];
    }
    if (threadIdx.x == kOffsetsBlock - 1) offsets[n_shards] = s_sum[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_compact(HitQ hq, PathQ q, PathQ q_new, float4* h0_new, float4* h1_new, const uint32_t* offsets) {
    const uint32_t shard = shard_index();
    const size_t sbase = size_t(shard) * kShard;
    const uint32_t n = hq.count[shard], first = offsets[shard];
    for (uint32_t j = threadIdx.x; j < n; j += 256u) {
        const float4 h0 = hq.h0[sbase + j], h1 = hq.h1[sbase + j];
        const size_t pi = sbase + as_u(h0.x);
        const uint32_t g = first + j;
        q_new.a[g] = q.a[pi];
        q_new.b[g] = q.b[pi];
        q_new.c[g] = q.c[pi];
        q_new.slot[g] = q.slot[pi];
        h0_new[g] = make_float4(as_f(g % kShard), h0.y, h0.z, h0.w);
        h1_new[g] = h1;
    }
}
// fills of the dense shards after k_compact: `*total` entries in shards of kShard
__global__ void k_dense_counts(const uint32_t* total, uint32_t* path_count, uint32_t* hit_count, uint32_t n_shards) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_shards) return;
    const uint32_t t = *total, lo = s * kShard;
    const uint32_t c = t > lo ? min(t - lo, kShard) : 0u;
    path_count[s] = c;
    hit_count[s] = c;
}

// ------------------------------------------------------------------------------------------------ film

// ImageBlock::put (src/block.cpp:60-95) for all samples of a pass.  A sample at block-local position p (the reference
// splats into 32x32 block images first, src/main.cpp:54-74, so positions and weights are taken relative to the block)
// adds value * wx * wy to the pixels [ceil(p-2), floor(p+2)]^2 with w = table[int(|pixel - p| * 16)].
//
// One workgroup owns a 16x16 pixel region (2x2 tiles, always inside one 32x32 block).  Per sample number the 256
// threads first publish their sample in LDS as {wx[4], wy[4], value*wx[4]} for the four film columns/rows it can reach
// with non-zero weight (pixel+1 .. pixel+4 in film coordinates; the table is 0 from distance 2 on), then every film
// pixel of the 20x20 patch gathers its up to 4x4 contributing samples into registers: no atomics in the loop, and a
// fixed summation order inside the region.  The patch is added to the film once at the end.
constexpr int kFilmRec = 20;     // floats per published sample: wy[4], wx[4], r*wx[4], g*wx[4], b*wx[4]
constexpr int kFilmStride = 21;  // odd record stride: neighbouring samples fall into different LDS banks (16-byte records read
                                 // with ds_read_b128 measured the same: the kernel moves LDS bytes, it is not short of issue slots)

__global__ __launch_bounds__(256) void k_film(DevScene sc, PassArgs pa, const float4* lbuf, float* film) {
    __shared__ float s_rec[256 * kFilmStride];
    const uint32_t regions_x = (pa.tiles_x + 1u) / 2u;
    const uint32_t rx = blockIdx.x % regions_x, ry = blockIdx.x / regions_x;
    const uint32_t sub = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t tx = 2u * rx + (sub & 1u), ty = 2u * ry + (sub >> 1);
    const uint32_t px = tx * 8u + (lane & 7u), py = ty * 8u + (lane >> 3);
    const bool in_image = tx < pa.tiles_x && ty < pa.tiles_y;
    // a region lies inside one 32x32 block, so either all of its tiles are rendered by this process or none
    const int32_t pos = !in_image ? -1 : (pa.tile_pos ? pa.tile_pos[ty * pa.tiles_x + tx] : int32_t(ty * pa.tiles_x + tx));
    if (__syncthreads_or(pos >= 0) == 0) return;
    const bool inside = pos >= 0 && px < sc.cam.width && py < sc.cam.height;
    const uint32_t tile = uint32_t(pos);
    // source pixel index inside the region: (sx, sy) in [0,16)^2
    const int sxp = int(px - 16u * rx), syp = int(py - 16u * ry);
    const int bx = int(px & ~31u), by = int(py & ~31u);
    const float bw = float(min(32, int(sc.cam.width) - bx)), bh = float(min(32, int(sc.cam.height) - by));
    float* my = s_rec + (syp * 16 + sxp) * kFilmStride;
    // what does not change from one sample number to the next: the pixel's Sobol' words, and the filter table in LDS (the
    // table is indexed per lane; as part of the kernel's arguments it was a memory gather per weight)
    __shared__ float s_filter[33];
    if (threadIdx.x < 33u) s_filter[threadIdx.x] = sc.filter[threadIdx.x];
    const uint2 xy = inside ? pixel_words(sc, px, py) : make_uint2(0u, 0u);
    __syncthreads();

    // the (up to) two patch pixels this thread accumulates: q0 = tid, q1 = tid + 256 (< 400)
    float acc[2][4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
    // one sample number: publish, then gather
    auto splat = [&](uint32_t s, const float4& L) {
        float rec[kFilmRec];
#pragma unroll
        for (int k = 0; k < kFilmRec; ++k) rec[k] = 0.0f;
        if (inside) {
            const V2 j = pixel_jitter(sc, pa.frame_tab + size_t(s) * 1024u, xy, px, py);
            const float qx = (float(int(px)) + j.x) - float(bx), qy = (float(int(py)) + j.y) - float(by);
            // ImageBlock::put drops a sample whose rounded position leaves its block (src/block.cpp:67-71); that happens
            // when pixel + jitter rounds up to the next integer in the last column or row of a block
            if (!(qx < 0.0f || qx >= bw || qy < 0.0f || qy >= bh)) {
                const float lx = qx + 2.0f, ly = qy + 2.0f;  // block-local film coordinates (2-pixel border)
                const int x0 = int(ceilf(lx - 2.0f)), x1 = int(floorf(lx + 2.0f));
                const int y0 = int(ceilf(ly - 2.0f)), y1 = int(floorf(ly + 2.0f));
                // film columns pixel+1 .. pixel+4 in block-local film coordinates
                const int cx = int(px) - bx + 1, cy = int(py) - by + 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int x = cx + k, y = cy + k;
                    const float wx = (x >= x0 && x <= x1) ? s_filter[min(int(fabsf(float(x) - lx) * 16.0f), 32)] : 0.0f;
                    const float wy = (y >= y0 && y <= y1) ? s_filter[min(int(fabsf(float(y) - ly) * 16.0f), 32)] : 0.0f;
                    rec[k] = wy;
                    rec[4 + k] = wx;
                    rec[8 + k] = L.x * wx;
                    rec[12 + k] = L.y * wx;
                    rec[16 + k] = L.z * wx;
                }
            }
        }
        __syncthreads();  // previous iteration's gathers are done
#pragma unroll
        for (int k = 0; k < kFilmRec; ++k) my[k] = rec[k];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int q = int(threadIdx.x) + 256 * r;
            if (q < 400) {
                const int u = q % 20, v = q / 20;  // patch pixel; patch origin = film coordinate 16 * region
                // a source pixel (sx, sy) reaches patch columns sx+1 .. sx+4, i.e. column u gets offset k = u - sx - 1
#pragma unroll
                for (int ky = 0; ky < 4; ++ky) {
                    const int sy = v - 1 - ky;
                    if (sy < 0 || sy > 15) continue;
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const int sx = u - 1 - kx;
                        if (sx < 0 || sx > 15) continue;
                        const float* src = s_rec + (sy * 16 + sx) * kFilmStride;
                        const float wy = src[ky];
                        acc[r][0] += src[8 + kx] * wy;   // (value * wx) * wy, as in src/block.cpp:90
                        acc[r][1] += src[12 + kx] * wy;
                        acc[r][2] += src[16 + kx] * wy;
                        acc[r][3] += src[4 + kx] * wy;   // weight channel: (1 * wx) * wy
                    }
                }
            }
        }
    };
    // four sample numbers at a time: their radiance loads are in flight together (64 bytes in a row per pixel in the
    // sample-major slot order)
    for (uint32_t s0 = 0; s0 < pa.spp_pass; s0 += 4u) {
        float4 Lq[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) Lq[k] = inside ? lbuf[sample_slot(pa, tile, lane, min(s0 + k, pa.spp_pass - 1u))] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            if (s0 + k < pa.spp_pass) splat(s0 + k, Lq[k]);
    }
    const int fw = int(sc.cam.width) + 4, fh = int(sc.cam.height) + 4;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int q = int(threadIdx.x) + 256 * r;
        if (q < 400) {
            const int fx = int(16u * rx) + q % 20, fy = int(16u * ry) + q / 20;
            if (fx < fw && fy < fh && acc[r][3] != 0.0f) {
                float* dst = film + (size_t(fy) * fw + fx) * 4;
                atomicAdd(dst + 0, acc[r][0]);
                atomicAdd(dst + 1, acc[r][1]);
                atomicAdd(dst + 2, acc[r][2]);
                atomicAdd(dst + 3, acc[r][3]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ sampler probe

__global__ void k_sobol(DevScene sc, const uint32_t* frame_tab, const int32_t* xy, const uint32_t* frame_local, const uint32_t* dim,
                        uint32_t n, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Sampler smp;
    smp.start(sc, frame_tab + size_t(frame_local[i]) * 1024u, uint32_t(xy[2 * size_t(i)]), uint32_t(xy[2 * size_t(i) + 1]), int(dim[i]));
    out[i] = smp.value(int(dim[i]));
}

// ------------------------------------------------------------------------------------------------ bare-function probes

// AABB::intersect (src/aabb.cpp:6-35) for explicit (box, ray) pairs, through the same routines and the same choice between
// the short division and the plain one as the reference-order traversal makes (closest_exact): accepted = the slabs
// overlap, the x/y entry lies before the ray's far end and the exit behind its near end.
__global__ void k_probe_box(DevScene sc, const float* boxes6, const float4* rays, uint32_t n, uint8_t* accepted) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 r0 = rays[2 * size_t(i)], r1 = rays[2 * size_t(i) + 1];
    const RayD ray{V3{r0.x, r0.y, r0.z}, V3{r0.w, r1.x, r1.y}, r1.z, r1.w};
    const RayPre pre = make_pre(sc, ray);
    const float* b = boxes6 + 6 * size_t(i);
    const BoxT t = pre.special ? box_decide<true>(b[0], b[1], b[2], b[3], b[4], b[5], pre) : box_decide<false>(b[0], b[1], b[2], b[3], b[4], b[5], pre);
    accepted[i] = (t.overlap && t.tn < ray.tmax) ? 1 : 0;
}

// Triangle::intersect (src/shape.cpp:50-89) for explicit (triangle, ray) pairs on the record form the leaf array holds
// ({p0, p1 - p0, p2 - p0}); tuv receives t, u, v of accepted hits (zeros otherwise).
// sin, cos of x[i] and x[i] ^ y[i] as the shading code computes them
__global__ void k_probe_math(const float* x, const float* y, uint32_t n, float* out3) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float sn, cs;
    sincos_f(x[i], sn, cs);
    out3[3 * size_t(i)] = sn;
    out3[3 * size_t(i) + 1] = cs;
    out3[3 * size_t(i) + 2] = pow_r(x[i], y[i]);
}

__global__ void k_probe_triangle(const float* tri9, const float4* rays, uint32_t n, uint8_t* hit, float* tuv) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 r0 = rays[2 * size_t(i)], r1 = rays[2 * size_t(i) + 1];
    const float* p = tri9 + 9 * size_t(i);
    const V3 p0{p[0], p[1], p[2]}, p1{p[3], p[4], p[5]}, p2{p[6], p[7], p[8]};
    float t = 0.0f, u = 0.0f, v = 0.0f;
    const bool h = tri_hit(p0, p1 - p0, p2 - p0, V3{r0.x, r0.y, r0.z}, V3{r0.w, r1.x, r1.y}, r1.z, r1.w, t, u, v);
    hit[i] = h ? 1 : 0;
    tuv[3 * size_t(i) + 0] = h ? t : 0.0f;
    tuv[3 * size_t(i) + 1] = h ? u : 0.0f;
    tuv[3 * size_t(i) + 2] = h ? v : 0.0f;
}

// ------------------------------------------------------------------------------------------------ division self test

// Feeds pseudo-random and adversarial operand pairs (mantissas of all ones, powers of two and their neighbours, slab-
// like magnitudes) to div_refine and counts disagreements with the division instruction sequence.
__global__ void k_div_selftest(uint64_t seed, uint32_t per_thread, unsigned long long* mismatches) {
    uint64_t x = seed ^ (0x9E3779B97F4A7C15ull * (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x + 1));
    auto next = [&]() {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        return x;
    };
    uint32_t bad = 0;
    for (uint32_t i = 0; i < per_thread; ++i) {
        const uint64_t r = next();
        uint32_t nb = uint32_t(r), db = uint32_t(r >> 32);
        // exponents: numerator 2^-60 .. 2^60, divisor 2^-30 .. 2^30 -- all that make_pre / box_decide let through to
        // div_refine (pairs outside that range, or with a quotient below kTinyQ, are skipped by div5_mismatch)
        const uint32_t ne = 67u + (nb >> 23) % 120u, de = 97u + (db >> 23) % 60u;
        nb = (nb & 0x807fffffu) | (ne << 23);
        db = (db & 0x807fffffu) | (de << 23);
        const uint32_t kind = uint32_t(next() % 8u);
        if (kind == 0u) db |= 0x007fffffu;                           // mantissa all ones
        if (kind == 1u) db &= 0xff800000u;                           // power of two
        if (kind == 2u) db = (db & 0xff800000u) | 1u;                // just above a power of two
        if (kind == 3u) nb |= 0x007fffffu;
        if (kind == 4u) nb = (nb & 0xff800000u) | (db & 0x007fffffu);  // equal mantissas
        bad += div5_mismatch(as_f(nb), as_f(db));
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

}  // namespace ptd
