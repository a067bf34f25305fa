// 3x3 convolution, stride 1, zero pad 1, NHWC, per-sample or shared weights, with
//   * the dense-block concat as a list of input views (no torch.cat copy),
//   * FBM modulation of the input formed in the loader (FDConv.py:456-479):
//       a = g1*x + (g2-g1)*L2 + (g3-g2)*L4 + (1-g3)*L8,
//   * nearest-2x upsampling folded into the loader (RRDBNet.forward),
//   * bias / LeakyReLU(0.2) / two scaled residuals in the epilogue.
//
// Two kernels:
//   conv3x3_mfma_bf16 : implicit GEMM on v_mfma_f32_32x32x16_bf16.  M = output channels (weights
//                       are the A operand), N = 32 consecutive pixels of one image row, K = 9 taps x
//                       32-channel chunks.  A workgroup (4 waves) owns a 16x32 pixel tile; the
//                       haloed 18x34x32ch input tile and the 9 x Cout x 32ch weight slab sit in LDS
//                       as 64-byte rows with a 16-byte-slot XOR swizzle so every ds_read_b128 of a
//                       fragment is conflict-free.  Each lane ends up with 4 consecutive output
//                       channels of one pixel per accumulator quad -> 8-byte NHWC stores.
//   conv3x3_generic   : fp32-accumulate VALU kernel for any dtype / channel count (fp32 parity
//                       path, 1- and 3-channel head and tail convs, odd-sized test networks).
#include "common.h"

namespace sr {

// ---------------------------------------------------------------------------------------------
// shared pieces
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float lrelu02(float v) { return v > 0.f ? v : 0.2f * v; }

template <typename T>
__device__ __forceinline__ float epilogue_one(float acc, int o, int b, long pix, const Epilogue& ep) {
    float y = acc + (ep.bias ? ep.bias[o] : 0.f);
    if (ep.slope) y = y > 0.f ? y : ep.slope[o] * y;
    else if (ep.lrelu) y = lrelu02(y);
    if (ep.r1.p) y = y * ep.s1 + Elem<T>::ld((const T*)ep.r1.p + voff(ep.r1, b, pix, o));
    if (ep.r2.p) y = y * ep.s2 + Elem<T>::ld((const T*)ep.r2.p + voff(ep.r2, b, pix, o));
    return y;
}

// ---------------------------------------------------------------------------------------------
// generic VALU kernel
// ---------------------------------------------------------------------------------------------
constexpr int CT = 16;   // tile side
constexpr int CCC = 8;   // channels per LDS stage

template <typename T, int OT>
__global__ __launch_bounds__(256) void conv3x3_generic(Concat in, int H, int W, const float* __restrict__ gates,
                                                       const T* __restrict__ wts, long w_bstride, long w_lo_off, int cin,
                                                       int cout, Epilogue ep, View out, int vec_out) {
    __shared__ float tile[CCC][CT + 2][CT + 3];
    __shared__ __attribute__((aligned(16))) float wsh[9][CCC][OT];
    const int tilesx = (W + CT - 1) / CT;
    const int b = blockIdx.y;
    const int o0 = blockIdx.z * OT;
    const int ty0 = (blockIdx.x / tilesx) * CT, tx0 = (blockIdx.x % tilesx) * CT;
    const int ty = threadIdx.x / CT, tx = threadIdx.x % CT;
    const int Ws = W >> in.ups;
    const T* wb = wts + (long)b * w_bstride;
    float acc[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) acc[o] = 0.f;
    for (int s = 0; s < in.nseg; ++s) {
        const View vx = in.x[s];
        for (int cc = 0; cc < vx.C; cc += CCC) {
            const int nc = min(CCC, vx.C - cc);
            __syncthreads();
            for (int it = threadIdx.x; it < (CT + 2) * (CT + 2) * nc; it += blockDim.x) {
                const int c = it % nc, pi = it / nc;
                const int py = pi / (CT + 2), px = pi % (CT + 2);
                const int h = ty0 + py - 1, w = tx0 + px - 1;
                float val = 0.f;
                if (h >= 0 && h < H && w >= 0 && w < W) {
                    val = Elem<T>::ld((const T*)vx.p + voff(vx, b, (long)(h >> in.ups) * Ws + (w >> in.ups), cc + c));
                    if (in.bands) {
                        const long p = (long)h * W + w;
                        const float4 g = ((const float4*)gates)[(long)b * H * W + p];
                        const float v2 = Elem<T>::ld((const T*)in.l2[s].p + voff(in.l2[s], b, p, cc + c));
                        const float v4 = Elem<T>::ld((const T*)in.l4[s].p + voff(in.l4[s], b, p, cc + c));
                        const float v8 = Elem<T>::ld((const T*)in.l8[s].p + voff(in.l8[s], b, p, cc + c));
                        val = g.x * val + (g.y - g.x) * v2 + (g.z - g.y) * v4 + (1.0f - g.z) * v8;
                    }
                }
                tile[c][py][px] = val;
            }
            for (int it = threadIdx.x; it < 9 * nc * OT; it += blockDim.x) {
                const int o = it % OT, c = (it / OT) % nc, tap = it / (OT * nc);
                float wv = 0.f;
                if (o0 + o < cout) {
                    const long at = ((long)tap * cout + o0 + o) * cin + in.cstart[s] + cc + c;
                    wv = Elem<T>::ld(wb + at);
                    if (w_lo_off) wv += Elem<T>::ld(wb + w_lo_off + at);  // split weights: hi + lo
                }
                wsh[tap][c][o] = wv;
            }
            __syncthreads();
            for (int c = 0; c < nc; ++c) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float x = tile[c][ty + tap / 3][tx + tap % 3];
#pragma unroll
                    for (int o = 0; o < OT; ++o) acc[o] += x * wsh[tap][c][o];
                }
            }
        }
    }
    const int h = ty0 + ty, w = tx0 + tx;
    if (h < H && w < W) {
        const long pix = (long)h * W + w;
        // bf16 outputs whose 8-channel groups are contiguous and 16-byte aligned (every tensor the executors allocate): one
        // 16-byte store per group instead of eight 2-byte ones (a thin-input layer such as SRVGGNetCompact's 3 -> 64 head is
        // bound by its stores: 2.56 -> 1.17 ms for four 1080p frames)
        if (sizeof(T) == 2 && OT >= 8 && vec_out) {
#pragma unroll
            for (int g = 0; g < OT / 8; ++g) {
                if (o0 + 8 * g + 7 < cout) {
                    uint32_t pk[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        pk[j] = pack2(epilogue_one<T>(acc[8 * g + 2 * j], o0 + 8 * g + 2 * j, b, pix, ep),
                                      epilogue_one<T>(acc[8 * g + 2 * j + 1], o0 + 8 * g + 2 * j + 1, b, pix, ep));
                    *(uint4*)((T*)out.p + voff(out, b, pix, o0 + 8 * g)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                } else {
#pragma unroll
                    for (int o = 8 * g; o < 8 * g + 8; ++o)
                        if (o0 + o < cout)
                            Elem<T>::st((T*)out.p + voff(out, b, pix, o0 + o), epilogue_one<T>(acc[o], o0 + o, b, pix, ep));
                }
            }
            return;
        }
#pragma unroll
        for (int o = 0; o < OT; ++o)
            if (o0 + o < cout)
                Elem<T>::st((T*)out.p + voff(out, b, pix, o0 + o), epilogue_one<T>(acc[o], o0 + o, b, pix, ep));
    }
}

template <typename T>
static int launch_generic(int B, int H, int W, const Concat& in, const float* gates, const void* wts, long wbs, long wlo,
                          int cout, const Epilogue& ep, View out, hipStream_t st) {
    const int tiles = cdiv(H, CT) * cdiv(W, CT);
    const int vec_out = sizeof(T) == 2 && view_vec_ok(out, 8) ? 1 : 0;
#define SR_GEN(OT)                                                                                                   \
    conv3x3_generic<T, OT><<<dim3(tiles, B, cdiv(cout, OT)), 256, 0, st>>>(in, H, W, gates, (const T*)wts, wbs, wlo, in.ctot, \
                                                                          cout, ep, out, vec_out)
    if (cout <= 4) SR_GEN(4);
    else if (cout <= 8) SR_GEN(8);
    else if (cout <= 16) SR_GEN(16);
    else SR_GEN(32);
#undef SR_GEN
    SR_LAUNCH_CHECK();
    return SR_OK;
}

// ---------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) short short8;
typedef __attribute__((ext_vector_type(16))) float float16v;

constexpr int MTH = 16, MTW = 32;            // pixel tile
constexpr int PH = MTH + 2, PW = MTW + 2;    // haloed tile
constexpr int NPIX = PH * PW;                // 612 rows of 64 B

// byte offset of 16-byte slot `chunk` (0..3) of 64-byte row `row`, XOR-swizzled so that the 16
// lanes of every ds_read_b128 lane group (rows distinct mod 16) land on 16 distinct slots.
__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// MT = number of 32-wide output-channel tiles handled by the workgroup (1 or 2).
//
// Producer / consumer workgroup of 8 waves, one per CU, two LDS stages:
//   waves 4-7 (loaders)  stream the next 32-channel chunk global -> registers -> LDS (forming the FBM
//                        modulated operand on the way), keeping tens of KiB of loads in flight per CU;
//   waves 0-3 (MFMA)     run the 9 taps x 2 k-steps of the current chunk out of the other stage.
// One barrier per chunk; the chunk time is max(HBM stream, MFMA), i.e. the kernel runs at whichever
// roofline the layer is on (the banded FDConv operand is 4 tensors wide -> HBM).
struct ChunkCursor {  // walks the virtual concat in 32-channel chunks
    int s, cc, kc;
};

// 16 bytes per lane global -> LDS without a register stop: LDS destination = wave-uniform `l` + lane * 16
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the wave's global loads
// (s_waitcnt vmcnt(0)): at the barrier in the middle of a chunk that would empty the loaders' two-deep load pipeline
// once more per chunk (measured: +20 % conv time).  The data handed over at that barrier is LDS data only.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ bool advance(const Concat& in, ChunkCursor& cur) {
    cur.cc += 32;
    cur.kc += 32;
    if (cur.cc >= in.x[cur.s].C) {
        cur.cc = 0;
        cur.s += 1;
    }
    return cur.s < in.nseg;
}

// Tile walk of one persistent workgroup.  Workgroups are dealt round-robin to the XCDs, so workgroup j lives on
// XCD j % 8; each XCD owns one contiguous range of the (tile, sample, cout block) index space and its workgroups
// stride through it together, so the tiles resident on an XCD at any time are neighbours (halo rows meet in L2).
struct TileWalk {
    unsigned next, end, stride;
    __device__ __forceinline__ TileWalk(unsigned total) {
        const unsigned j = blockIdx.x, x = j & 7u, q = total >> 3, r = total & 7u;
        const unsigned start = x * q + (x < r ? x : r);
        end = start + q + (x < r ? 1u : 0u);
        // workgroups on this XCD: ceil((gridDim - x) / 8)
        stride = (gridDim.x - x + 7u) >> 3;
        next = start + (j >> 3);
    }
};

struct TileId {
    int b, o_blk, h0, w0;
};
template <int MT>
__device__ __forceinline__ TileId decode_tile(unsigned lin, unsigned tiles, int tilesx, int B) {
    TileId t;
    const int tile = lin % tiles;
    lin /= tiles;
    t.b = lin % (unsigned)B;
    t.o_blk = (lin / (unsigned)B) * (MT * 32);
    t.h0 = (tile / tilesx) * MTH;
    t.w0 = (tile % tilesx) * MTW;
    return t;
}

// SPLIT != 0: wts holds the high parts, wts + w_lo_off the low parts of the same [B][9][cout][cin] weights (w = hi + lo, both
// bf16: the rounding of the dynamic weights, an error coherent over the whole image, was the dominant bf16 error of a deep
// network -- DESIGN.md 6, finding 15); the MFMA waves then issue two MFMAs per pixel fragment, one per part.
// LDS: two pixel stages, then the slabs:  SPLIT 0: hi of chunk parity 0 / 1;  SPLIT 2 (MT = 1): [hi][lo] per chunk parity;
// SPLIT 1 (MT = 2): one [hi][lo] pair.  Barrier protocol (every wave executes the same number of workgroup barriers):
//   SPLIT 0, 2: loaders  pixels(s) -> barrier_s                          MFMA waves  barrier_s -> DMA slabs(s+1) -> multiply(s)
//   SPLIT 1   : loaders  pixels(s) first half -> barrier'_{s-1} -> second half -> barrier_s  (+ one barrier after the last chunk)
//               MFMA waves  barrier_s -> multiply(s) -> barrier'_s -> DMA slabs(s+1)
// RES (shared weights, SPLIT = 0, one output block): the slabs of EVERY chunk of the layer stay in LDS for the whole kernel
// (a plain 64 -> 64 layer is 73.7 KB, as much as the double-buffered slabs of the streaming form): they are fetched once
// per workgroup instead of once per tile and chunk -- for a plain layer the slab is as many bytes as the chunk's pixels --
// and no chunk barrier has to drain the MFMA waves' vector-memory counter any more, so the epilogue's stores retire
// behind the next tile's multiplies.
// F16 (BANDS, SPLIT = 0; opt-in, DESIGN.md finding 39): the weights are ONE fp16 value each, the loader packs the mixed operand to
// fp16 and the multiply is v_mfma_f32_32x32x16_f16 -- the instruction, slab and fragment counts of unsplit weights.
template <int MT, bool BANDS, int SPLIT, bool RES = false, bool F16 = false>
__global__ __launch_bounds__(512) void conv3x3_mfma_bf16(Concat in, int B, int H, int W, const float* __restrict__ gates,
                                                         const bf16_t* __restrict__ wts, long w_bstride, long w_lo_off,
                                                         int cin, int cout, Epilogue ep, View out, unsigned total, int wmap) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int STAGE_PX = NPIX * 64;
    // F16: a tenth weight plane per chunk -- the 9-tap SUM of what rounding the weights to fp16 lost -- multiplies the centre pixel
    // once more: the part of the rounding error that is coherent over the image (a gain error of the layer) is taken out for a
    // ninth of the second pass a hi + lo pair costs (DESIGN.md finding 39)
    constexpr int TAPS = F16 ? 10 : 9;
    constexpr int WSLAB = TAPS * MT * 32 * 64;
    char* const lds_wa = lds + 2 * STAGE_PX;
    char* const lds_wb = lds_wa + WSLAB;
    const int tilesx = (W + MTW - 1) / MTW;
    const unsigned tiles = (unsigned)tilesx * ((H + MTH - 1) / MTH);
    // wmap (A/B hook): physical waves {0, 1, 4, 5} multiply and {2, 3, 6, 7} load, so that -- waves being dealt to the four
    // SIMDs round-robin -- two SIMDs hold the MFMA waves and the other two the loaders
    const int lane = threadIdx.x & 63;
    const int pw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wmap ? ((pw & 2) * 2 + (pw & 1) + ((pw >> 2) << 1)) : pw;  // scalar
    const int tid = wave * 64 + lane;
    const bool loader = wave >= 4;
    const int r = lane & 31, hk = lane >> 5;
    const int nchunks = cin / 32;
    TileWalk walk(total);
    if (walk.next >= walk.end) return;  // whole workgroup: no barrier is skipped by a subset of waves

    if (loader) {
        // ---- producer waves: never touch the accumulators (separate register budget).  They run one chunk ahead of the
        // MFMA waves across tile boundaries, so a tile's first chunk streams in while the previous tile is still being
        // multiplied and written out.
        //
        // A chunk = the haloed 18 x 34 pixel tile of 32 channels (612 pixels x 4 slots of 8 channels) + its weight slab(s).
        // Pixels move in ROUNDS of G items per thread, software-pipelined two deep ACROSS chunk boundaries: while round r
        // is modulated and written to LDS, round r + 1 -- for the last round of a chunk: round 0 of the NEXT chunk, and the
        // next chunk's weight slabs -- is already in flight.  The loads land in registers, which nobody else sees, so they
        // may be in flight across the workgroup barrier; the barrier here therefore waits for this wave's LDS traffic only
        // (lds_barrier; __syncthreads() would drain vmcnt and expose one full HBM latency per chunk: measured 21.6k
        // of a chunk's 25.5k cycles in the loader with the pipeline restarted per chunk).
        // The loaders move pixels only; the weight slabs are fetched by the MFMA waves themselves (LDS-DMA, below).
        const int ltid = tid - 256;
        // the loaders are the pole of the workgroup and, as waves 4-7, lose the age-based issue arbitration against their
        // SIMD partners: static priority for them (A/B: -3.5 % on the Cout = 32 layers, nothing at Cout = 64)
        __builtin_amdgcn_s_setprio(2);
        constexpr int G = BANDS ? 2 : 5;  // (three items per round in the fp16 Cout = 32 kernel, 237 VGPRs, 192 B in flight per lane: +5 % conv time)
        constexpr int ROUNDS = (NPIX * 4 + 256 * G - 1) / (256 * G);
        // SPLIT = 1: rounds ahead of the barrier at which the MFMA waves hand the slab pair back (late, so that the loaders
        // rarely wait for the fused multiply; measured equal to the half-way position: the MFMA waves are the pole there)
        constexpr int RA = ROUNDS - 1;
        struct Batch {
            uint4 vx[G], a2[G], a4[G], a8[G];
            float4 gg[G];
        };
        struct Desc {  // one chunk of one tile
            int b, h0, w0, o_blk, s, cc, kc;
        };
        const int Ws = W >> in.ups;
        // Addressing is split by how often it changes (the loaders' own instruction issue was a measured third of a
        // chunk's time; divisions and 64-bit address arithmetic per item and chunk made up half of it):
        //   per kernel  item k of this thread -> 8-channel slot and LDS offset inside the haloed tile
        //               (two layouts: slots of a pixel adjacent, or -- blocks of 8 channels -- pixels of a slot adjacent);
        //   per tile    clamped image row / column of the items and the inside-image bits;
        //   per chunk   one uniform base pointer per tensor and, per thread, the element offset of its slot's channels;
        //   per item    byte offset = pixel offset * pixel stride + slot offset: one v_mad per tensor, added to the
        //               uniform base by the load itself.
        constexpr int NIT = ROUNDS * G;
        unsigned it_lds[NIT];  // swz(pi, slot) | slot << 16
        unsigned it_hw[NIT];   // clamped (row << 16 | column) of the item in the current tile
        unsigned inside = 0;   // bit k: item k lies inside the image
        int geo_kind = -1;
        auto item_pos = [&](int kind, int k, int& slot, int& pi) {
            const int it = ltid + k * 256;
            if (kind) {
                slot = it / NPIX;
                pi = it - slot * NPIX;
            } else {
                slot = it & 3;
                pi = it >> 2;
            }
            slot &= 3;
        };
        auto build_geo = [&](int kind) {
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                int slot, pi;
                item_pos(kind, k, slot, pi);
                it_lds[k] = (unsigned)swz(pi, slot) | ((unsigned)slot << 16);
            }
            geo_kind = kind;
        };
        auto build_tile = [&](const Desc& d) {
            inside = 0;
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                int slot, pi;
                item_pos(geo_kind, k, slot, pi);
                const int py = pi / PW, px = pi - py * PW;
                const int h = d.h0 + py - 1, w = d.w0 + px - 1;
                const int hc = min(max(h, 0), H - 1), wc = min(max(w, 0), W - 1);
                it_hw[k] = ((unsigned)hc << 16) | (unsigned)wc;
                if (h == hc && w == wc) inside |= 1u << k;
            }
        };
        // element offset of the thread's 8-channel slot `q` of chunk d inside view v (sample and pixel terms excluded)
        auto slot_off = [&](const View& v, const Desc& d, unsigned q) {
            const unsigned cc = (unsigned)(v.c0 + d.cc) + 8u * q;
            return (cc >> v.csh) * (unsigned)v.cbs + (cc & ((1u << v.csh) - 1u));
        };
        // every load is issued unconditionally from a clamped (always valid) address and zeroed afterwards: a branch around
        // a load makes hipcc drain the memory pipeline per element
        auto issue = [&](const Desc& d, int r, Batch& bt) {
            const View vx = in.x[d.s];
            const char* xb = (const char*)((const bf16_t*)vx.p + (long)d.b * vx.bs);
            const View v2 = in.l2[d.s], v4 = in.l4[d.s], v8 = in.l8[d.s];
            const char* b2 = BANDS ? (const char*)((const bf16_t*)v2.p + (long)d.b * v2.bs) : nullptr;
            const char* b4 = BANDS ? (const char*)((const bf16_t*)v4.p + (long)d.b * v4.bs) : nullptr;
            const char* b8 = BANDS ? (const char*)((const bf16_t*)v8.p + (long)d.b * v8.bs) : nullptr;
            const char* gb = (const char*)((const float4*)gates + (long)d.b * H * W);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int k = r * G + g;
                const unsigned q = it_lds[k] >> 16;
                const unsigned hc = it_hw[k] >> 16, wc = it_hw[k] & 0xffffu;
                const unsigned p = hc * (unsigned)W + wc;
                const unsigned ps = in.ups ? (hc >> 1) * (unsigned)Ws + (wc >> 1) : p;
                bt.vx[g] = *(const uint4*)(xb + (size_t)(2u * (ps * (unsigned)vx.ld + slot_off(vx, d, q))));
                if (BANDS) {
                    bt.a2[g] = *(const uint4*)(b2 + (size_t)(2u * (p * (unsigned)v2.ld + slot_off(v2, d, q))));
                    bt.a4[g] = *(const uint4*)(b4 + (size_t)(2u * (p * (unsigned)v4.ld + slot_off(v4, d, q))));
                    bt.a8[g] = *(const uint4*)(b8 + (size_t)(2u * (p * (unsigned)v8.ld + slot_off(v8, d, q))));
                    bt.gg[g] = *(const float4*)(gb + (size_t)(p << 4));
                }
            }
        };
        auto consume = [&](int r, const Batch& bt, char* lds_px) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int k = r * G + g;
                uint4 v = bt.vx[g];
                if (BANDS) v = modulate8t<F16>(v, bt.a2[g], bt.a4[g], bt.a8[g], bt.gg[g].x, bt.gg[g].y - bt.gg[g].x, bt.gg[g].z - bt.gg[g].y, 1.0f - bt.gg[g].z);
                if (!((inside >> k) & 1u)) v = make_uint4(0u, 0u, 0u, 0u);
                if ((k + 1) * 256 <= NPIX * 4 || ltid + k * 256 < NPIX * 4) *(uint4*)(lds_px + (it_lds[k] & 0xffffu)) = v;
            }
        };
        auto make_desc = [&](unsigned lin) {
            const TileId t = decode_tile<MT>(lin, tiles, tilesx, B);
            return Desc{t.b, t.h0, t.w0, t.o_blk, 0, 0, 0};
        };
        unsigned lin = walk.next;
        Desc d = make_desc(lin);
        Batch pre, oth;
        build_geo(in.x[0].ld == 8);
        build_tile(d);
        issue(d, 0, pre);
        unsigned step = 0;
        bool more = true;
        while (more) {
            // the chunk after this one (this tile's next 32 channels, or the first chunk of the workgroup's next tile);
            // past the end: this chunk again, as a valid address for loads nobody consumes
            Desc dn = d;
            dn.cc += 32;
            dn.kc += 32;
            if (dn.cc >= in.x[dn.s].C) {
                dn.cc = 0;
                dn.s += 1;
            }
            if (dn.s >= in.nseg) {
                lin += walk.stride;
                more = lin < walk.end;
                dn = more ? make_desc(lin) : d;
            }
            char* px = lds + (step & 1) * STAGE_PX;
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                if (SPLIT == 1 && r == RA && step) lds_barrier();  // the consumers have finished the previous chunk (see below)
                Batch& cur_b = (r & 1) ? oth : pre;
                Batch& nxt_b = (r & 1) ? pre : oth;
                if (r + 1 < ROUNDS) {
                    issue(d, r + 1, nxt_b);
                    consume(r, cur_b, px);
                } else {
                    // last round: this chunk's loads are all in registers; retarget the tables, then start the next chunk
                    const int kind_n = in.x[dn.s].ld == 8;
                    if (kind_n != geo_kind) {  // segments of different layouts in one concat (rare): no overlap here
                        consume(r, cur_b, px);
                        build_geo(kind_n);
                        build_tile(dn);
                        issue(dn, 0, nxt_b);
                    } else {
                        const unsigned inside_d = inside;
                        if (dn.b != d.b || dn.h0 != d.h0 || dn.w0 != d.w0) build_tile(dn);
                        issue(dn, 0, nxt_b);
                        const unsigned inside_n = inside;
                        inside = inside_d;
                        consume(r, cur_b, px);
                        inside = inside_n;
                    }
                }
            }
            if (ROUNDS & 1) pre = oth;  // the next chunk's round 0 sits in `oth` after an odd number of rounds
            lds_barrier();  // pixel stage of chunk `step` is in place
            d = dn;
            ++step;
        }
        if (SPLIT == 1) lds_barrier();  // partner of the consumers' barrier after the very last chunk
        return;
    }
    unsigned step = 0;
    float* bias_lds = (float*)(lds + 2 * STAGE_PX + (RES ? nchunks : (SPLIT == 2 ? 4 : 2)) * WSLAB);  // [cout], zeros without a bias
    float* slope_lds = bias_lds + cout;  // [cout] negative slopes: PReLU's, 0.2 for LeakyReLU, 1 without activation
    for (int i = tid; i < cout; i += 256) {
        bias_lds[i] = ep.bias ? ep.bias[i] : 0.f;
        slope_lds[i] = ep.slope ? ep.slope[i] : (ep.lrelu ? 0.2f : 1.0f);
    }
    // Weight slabs: rows (tap, o) x 32 channels of [B][9][cout][cin], L2-resident, copied global -> LDS by the MFMA waves
    // with global_load_lds_dwordx4 (no registers, no ds_write; the loaders -- the busier half of the workgroup -- carried
    // them through 36-72 registers before and spilled).  One instruction writes 1 KiB = 16 slab rows in lane order, so
    // the XOR swizzle of the LDS image is applied to the SOURCE address: lane l fills 16-byte slot l & 3 of row l >> 2,
    // which holds channel group (l & 3) ^ ((row >> 2) & 3).  Rows of a thin layer beyond cout repeat the last row: an
    // output row depends on its own weight row only and the epilogue never stores those rows.
    auto slab_dma = [&](int sb, int so_blk, int skc, long part_off, char* lds_slab) {
        constexpr int NI = TAPS * MT * 32 / 16;  // 1 KiB pieces per slab
        const bf16_t* base = wts + (long)sb * w_bstride + part_off;
#pragma unroll
        for (int jj = 0; jj < (NI + 3) / 4; ++jj) {
            const int j = jj * 4 + wave;
            if (j < NI) {
                const int row = j * 16 + (lane >> 2);
                const int cgrp = (lane & 3) ^ ((row >> 2) & 3);
                const int tap = row / (MT * 32), o = row - tap * (MT * 32);
                const int oc = min(so_blk + o, cout - 1);
                glds16(base + ((long)tap * cout + oc) * cin + skc + cgrp * 8, lds_slab + j * 1024);
            }
        }
    };
    if (RES) {  // every chunk's slab, once (shared weights: sample 0, output block 0)
        for (int c = 0; c < nchunks; ++c) slab_dma(0, 0, c * 32, 0, lds_wa + c * WSLAB);
    } else {  // slabs of the workgroup's very first chunk
        const TileId t0 = decode_tile<MT>(walk.next, tiles, tilesx, B);
        if (SPLIT == 2) {
            slab_dma(t0.b, t0.o_blk, 0, 0, lds_wa);
            slab_dma(t0.b, t0.o_blk, 0, w_lo_off, lds_wa + WSLAB);
        } else {
            slab_dma(t0.b, t0.o_blk, 0, 0, lds_wa);
            if (SPLIT == 1) slab_dma(t0.b, t0.o_blk, 0, w_lo_off, lds_wb);
        }
    }
    for (unsigned lin = walk.next; lin < walk.end; lin += walk.stride) {
        const TileId t = decode_tile<MT>(lin, tiles, tilesx, B);
        const int b = t.b, o_blk = t.o_blk, h0 = t.h0, w0 = t.w0;
        float16v acc[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
        // One k-step = (tap, half of the 32-channel chunk): MT weight fragments (x2 with fused split weights) and 4 pixel
        // fragments feed 4*MT (x2) MFMAs.  The fragments of step s + 1 are requested BEFORE the MFMAs of step s are
        // issued (two register sets): left to itself the compiler reads one pixel fragment, waits for it, issues the two
        // MFMAs that use it and only then computes the next address -- every pair of MFMAs then pays a full LDS latency
        // (measured: ~100 cycles per MFMA instead of 32; the second half-step of a split MT = 2 chunk cost +25 %).
        struct Frags {
            short8 a[MT], a2[MT], b[4];
        };
        auto load_step_tap = [&](int tap, int dy, int dx, int kk, Frags& f, const char* cur_px, const char* cur_w, const char* cur_w2) {
            const int chunk = kk * 2 + hk;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int wo = swz(tap * (MT * 32) + m * 32 + r, chunk);
                f.a[m] = *(const short8*)(cur_w + wo);
                if (SPLIT) f.a2[m] = *(const short8*)(cur_w2 + wo);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int pi = (wave * 4 + n + dy) * PW + r + dx;
                f.b[n] = *(const short8*)(cur_px + swz(pi, chunk));
            }
        };
        auto load_step = [&](int dy, int dx, int kk, Frags& f, const char* cur_px, const char* cur_w, const char* cur_w2) {
            load_step_tap(dy * 3 + dx, dy, dx, kk, f, cur_px, cur_w, cur_w2);
        };
        auto mfma_step = [&](const Frags& f) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    if constexpr (F16)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8v, f.a[m]), __builtin_bit_cast(half8v, f.b[n]), acc[m][n], 0, 0, 0);
                    else
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[m], f.b[n], acc[m][n], 0, 0, 0);
                    if (SPLIT) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a2[m], f.b[n], acc[m][n], 0, 0, 0);
                }
        };
        auto multiply = [&](const char* cur_px, const char* cur_w, const char* cur_w2) {
            Frags f0, f1;
            load_step(0, 0, 0, f0, cur_px, cur_w, cur_w2);
            // dy stays a real loop: fully unrolled, the 36 swizzled pixel addresses are hoisted into registers and
            // (with 128 accumulators at MT = 2) spill
#pragma unroll 1
            for (int dy = 0; dy < 3; ++dy) {
                load_step(dy, 0, 1, f1, cur_px, cur_w, cur_w2);
                mfma_step(f0);
                load_step(dy, 1, 0, f0, cur_px, cur_w, cur_w2);
                mfma_step(f1);
                load_step(dy, 1, 1, f1, cur_px, cur_w, cur_w2);
                mfma_step(f0);
                load_step(dy, 2, 0, f0, cur_px, cur_w, cur_w2);
                mfma_step(f1);
                load_step(dy, 2, 1, f1, cur_px, cur_w, cur_w2);
                mfma_step(f0);
                load_step(dy < 2 ? dy + 1 : 2, 0, 0, f0, cur_px, cur_w, cur_w2);  // last round: a harmless re-read, no branch
                mfma_step(f1);
            }
            if constexpr (F16) {  // plane 9 (the summed rounding remainders) times the centre pixel
                load_step_tap(9, 1, 1, 0, f0, cur_px, cur_w, cur_w2);
                load_step_tap(9, 1, 1, 1, f1, cur_px, cur_w, cur_w2);
                mfma_step(f0);
                mfma_step(f1);
            }
        };
        for (int c = 0; c < nchunks; ++c) {
            const char* cur_px = lds + (step & 1) * STAGE_PX;
            // where the NEXT chunk's weights come from (this tile's next 32 channels or the next tile's first chunk)
            const bool last_c = c + 1 == nchunks;
            const bool have_next = !last_c || lin + walk.stride < walk.end;
            int nb = b, no = o_blk;
            if (last_c && have_next) {
                const TileId tn = decode_tile<MT>(lin + walk.stride, tiles, tilesx, B);
                nb = tn.b;
                no = tn.o_blk;
            }
            const int nkc = last_c ? 0 : (c + 1) * 32;
            if (RES && step) lds_barrier();  // resident slabs: nothing of this wave's vector memory traffic is waited for
            else __syncthreads();  // chunk `step`: pixel stage in place (loaders), own slab DMA drained (vmcnt), previous chunk consumed
            if (RES) {
                multiply(cur_px, lds_wa + c * WSLAB, nullptr);
            } else if (SPLIT == 1) {
                // one slab pair [hi][lo] (four slabs of 36 KB do not fit next to the two pixel stages at MT = 2): the next
                // chunk's slabs can only be requested once every MFMA wave has finished this chunk -- second barrier
                multiply(cur_px, lds_wa, lds_wb);
                lds_barrier();
                if (have_next) {
                    slab_dma(nb, no, nkc, 0, lds_wa);
                    slab_dma(nb, no, nkc, w_lo_off, lds_wb);
                }
            } else if (SPLIT == 2) {
                char* nslab = lds_wa + ((step + 1) & 1) * 2 * WSLAB;
                if (have_next) {
                    slab_dma(nb, no, nkc, 0, nslab);
                    slab_dma(nb, no, nkc, w_lo_off, nslab + WSLAB);
                }
                const char* slab = lds_wa + (step & 1) * 2 * WSLAB;
                multiply(cur_px, slab, slab + WSLAB);
            } else {
                if (have_next) slab_dma(nb, no, nkc, 0, ((step + 1) & 1) ? lds_wb : lds_wa);
                multiply(cur_px, (step & 1) ? lds_wb : lds_wa, nullptr);
            }
            ++step;
        }
        // ---- epilogue: lane (r, hk) holds, for pixel (h, w0 + r), channels 8q + 4hk + {0..3} of each 32-channel tile.
        // v_permlane32_swap between lanes r and r + 32 regroups them into 16 consecutive channels per lane (low lanes
        // 0..15, high lanes 16..31), so residuals are read and results written as 16-byte pieces, two per lane,
        // instead of four 8-byte ones (the 8-byte version took 6 us per tile on its own).
        const int w = w0 + r;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int h = h0 + wave * 4 + n;
            const bool inside = h < H && w < W;
            const long pix = (long)min(h, H - 1) * W + min(w, W - 1);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float y[16];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[m][n][j]), __float_as_uint(acc[m][n][8 + j]), false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[m][n][4 + j]), __float_as_uint(acc[m][n][12 + j]), false, false);
                    y[j] = __uint_as_float(s0[0]);
                    y[4 + j] = __uint_as_float(s0[1]);
                    y[8 + j] = __uint_as_float(s1[0]);
                    y[12 + j] = __uint_as_float(s1[1]);
                }
                if (cout < 32) {  // thin tail (conv_last: 1 or 3 channels): channel-wise epilogue from the low lanes
                    if (inside && hk == 0 && m == 0) {
#pragma unroll
                        for (int j = 0; j < 16; ++j)
                            if (j < cout)
                                Elem<bf16_t>::st((bf16_t*)out.p + voff(out, b, pix, j), epilogue_one<bf16_t>(y[j], j, b, pix, ep));
                    }
                    continue;
                }
                // bias (staged in LDS once per workgroup) + activation, written back over the accumulators
                const float4* bl = (const float4*)(bias_lds + o_blk + m * 32 + hk * 16);  // 16-byte aligned: four reads, not 16
                const float4* sl = (const float4*)(slope_lds + o_blk + m * 32 + hk * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bb = bl[q], ss = sl[q];
                    const float bq[4] = {bb.x, bb.y, bb.z, bb.w}, sq[4] = {ss.x, ss.y, ss.z, ss.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = y[4 * q + j] + bq[j];
                        acc[m][n][4 * q + j] = v > 0.f ? v : sq[j] * v;
                    }
                }
            }
        }
        if (cout < 32) continue;
        // Residuals: all loads of one residual are issued together and folded into the accumulators before the first
        // store of the tile goes out.  (Loading them piecewise between the stores -- and the bias from global memory
        // per 32x32 block -- made every piece wait for the stores before it to drain: 8 us per tile without and 21 us
        // with residuals at Cout = 64, none of it hidden behind the loaders.)
        auto fold = [&](const View& rv, float sc) {
            uint4 rr[MT][4][2];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const long pix = (long)min(h0 + wave * 4 + n, H - 1) * W + min(w, W - 1);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int half = 0; half < 2; ++half)
                        rr[m][n][half] = *(const uint4*)((const bf16_t*)rv.p + voff(rv, b, pix, o_blk + m * 32 + hk * 16 + 8 * half));
            }
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const uint32_t rw[4] = {rr[m][n][half].x, rr[m][n][half].y, rr[m][n][half].z, rr[m][n][half].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[m][n][8 * half + 2 * j] = acc[m][n][8 * half + 2 * j] * sc + lo_f(rw[j]);
                            acc[m][n][8 * half + 2 * j + 1] = acc[m][n][8 * half + 2 * j + 1] * sc + hi_f(rw[j]);
                        }
                    }
        };
        if (ep.r1.p) fold(ep.r1, ep.s1);
        if (ep.r2.p) fold(ep.r2, ep.s2);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int h = h0 + wave * 4 + n;
            if (h >= H || w >= W) continue;
            const long pix = (long)h * W + w;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int o = 8 * half;
                    uint4 st;
                    st.x = pack2(acc[m][n][o], acc[m][n][o + 1]);
                    st.y = pack2(acc[m][n][o + 2], acc[m][n][o + 3]);
                    st.z = pack2(acc[m][n][o + 4], acc[m][n][o + 5]);
                    st.w = pack2(acc[m][n][o + 6], acc[m][n][o + 7]);
                    *(uint4*)((bf16_t*)out.p + voff(out, b, pix, o_blk + m * 32 + hk * 16 + 8 * half)) = st;
                }
        }
    }
}

int g_split_two_pass = 0;  // A/B hook (sr_debug_split_two_pass): experiments in tools/bench_ops.py

static bool mfma_eligible(const Concat& in, int cout, const Epilogue& ep, const View& out, const void* wts,
                          long wbs) {
    const bool thin = cout <= 16;  // HBM-bound anyway: the unused output rows of the 32-wide tile cost nothing that matters
    if (cout % 32 != 0 && !thin) return false;
    if ((uintptr_t)wts % 16 != 0 || wbs % 8 != 0) return false;
    for (int s = 0; s < in.nseg; ++s) {
        if (in.x[s].C % 32 != 0 || !view_vec_ok(in.x[s], 8) || in.x[s].bs >= (1L << 31)) return false;
        if (in.bands) {
            const View* bv[3] = {&in.l2[s], &in.l4[s], &in.l8[s]};
            for (const View* v : bv)
                if (!view_vec_ok(*v, 8) || v->bs >= (1L << 31)) return false;
        }
    }
    if (thin) return out.p != nullptr;
    if (!view_vec_ok(out, 8)) return false;  // the epilogue moves 8-channel (16-byte) pieces
    if (ep.r1.p && !view_vec_ok(ep.r1, 8)) return false;
    if (ep.r2.p && !view_vec_ok(ep.r2, 8)) return false;
    return true;
}

template <int MT, bool BANDS, int SPLIT, bool RES = false, bool F16 = false>
static int launch_mfma(int B, int H, int W, const Concat& in, const float* gates, const void* wts, long wbs, long wlo, int cout,
                       const Epilogue& ep, View out, hipStream_t st) {
    const size_t nslab = RES ? (size_t)(in.ctot / 32) : (SPLIT == 2 ? 4 : 2);
    const size_t ldsb = 2 * (size_t)NPIX * 64 + nslab * (F16 ? 10ul : 9ul) * MT * 32 * 64 + 2 * (size_t)cout * sizeof(float);  // stages + slabs + bias + slopes
    if (ldsb > 160 * 1024) return SR_NOT_ELIGIBLE;
    if (RES && (wbs != 0 || wlo != 0 || cdiv(cout, MT * 32) != 1)) return SR_NOT_ELIGIBLE;
    static_assert(!F16 || (BANDS && SPLIT == 0 && !RES), "fp16 operands: banded layers with single weights");
    if (F16) wlo = 0;  // the marker, not an offset
    SR_LDS_ATTR(ldsb, conv3x3_mfma_bf16<MT, BANDS, SPLIT, RES, F16>);  // per device
    const long nblk = (long)cdiv(H, MTH) * cdiv(W, MTW) * B * cdiv(cout, MT * 32);
    if (nblk >= (1L << 31)) return SR_ERR_ARG;
    int ncu = 0;  // persistent workgroups: one per CU (the two LDS stages fill a CU's LDS)
    if (int rc = device_cus(&ncu)) return rc;
    const unsigned grid = (unsigned)(nblk < ncu ? nblk : ncu);
    conv3x3_mfma_bf16<MT, BANDS, SPLIT, RES, F16><<<dim3(grid), 512, ldsb, st>>>(in, B, H, W, gates, (const bf16_t*)wts, wbs, wlo, in.ctot, cout, ep, out, (unsigned)nblk, g_split_two_pass);
    SR_LAUNCH_CHECK();
    return SR_OK;
}

int g_force_generic = 0;  // test hook: route bf16 through the VALU kernel
int g_no_resident = 0;    // test hook (A/B): shared weights streamed per chunk like per-sample ones

bool conv3x3_f16_ok(int dtype, const Concat& in, const void* wts, long wbs, int cout, const Epilogue& ep, View out) {
    return dtype == SR_BF16 && in.bands && !g_force_generic && mfma_eligible(in, cout, ep, out, wts, wbs);
}

int conv3x3_launch(int dtype, int B, int H, int W, const Concat& in, const float* gates, const void* wts, long wbs,
                   long wlo, int cout, const Epilogue& ep, View out, hipStream_t st) {
    if (wlo == SR_WLO_F16) {  // one fp16 weight per MAC, operand packed to fp16 by the band-mixing loader: matrix-core kernel only
        if (dtype != SR_BF16 || !in.bands || g_force_generic || !mfma_eligible(in, cout, ep, out, wts, wbs)) return SR_ERR_ARG;
        return cout % 64 == 0 ? launch_mfma<2, true, 0, false, true>(B, H, W, in, gates, wts, wbs, wlo, cout, ep, out, st)
                              : launch_mfma<1, true, 0, false, true>(B, H, W, in, gates, wts, wbs, wlo, cout, ep, out, st);
    }
    if (dtype == SR_F32) return launch_generic<float>(B, H, W, in, gates, wts, wbs, wlo, cout, ep, out, st);
    if (!g_force_generic && mfma_eligible(in, cout, ep, out, wts, wbs)) {
        const bool two = (cout % 64 == 0);
        if (wlo % 8 != 0) return SR_ERR_ARG;
        {  // plain layers with LDS-resident weights: the wave-autonomous kernel of conv_plain.hip
            const int rcp = conv3x3_plain_launch(B, H, W, in, wts, wbs, wlo, cout, ep, out, st);
            if (rcp != SR_NOT_ELIGIBLE) return rcp;
        }
#define SR_MFMA(MT, BANDS, SPLIT) launch_mfma<MT, BANDS, SPLIT>(B, H, W, in, gates, wts, wbs, wlo, cout, ep, out, st)
        int rc;
        if (in.bands) {
            if (wlo) rc = two ? SR_MFMA(2, true, 1) : SR_MFMA(1, true, 2);
            else rc = two ? SR_MFMA(2, true, 0) : SR_MFMA(1, true, 0);
        } else {
            if (wlo) rc = two ? SR_MFMA(2, false, 1) : SR_MFMA(1, false, 2);
            else {
                // shared single-part weights that fit LDS whole (e.g. 64 -> 64: 73.7 KB): resident slabs
                rc = SR_NOT_ELIGIBLE;
                if (wbs == 0 && !g_no_resident)
                    rc = two ? launch_mfma<2, false, 0, true>(B, H, W, in, gates, wts, wbs, wlo, cout, ep, out, st)
                             : launch_mfma<1, false, 0, true>(B, H, W, in, gates, wts, wbs, wlo, cout, ep, out, st);
                if (rc == SR_NOT_ELIGIBLE) rc = two ? SR_MFMA(2, false, 0) : SR_MFMA(1, false, 0);
            }
        }
#undef SR_MFMA
        // a layer whose bias / slope tables push the LDS image past 160 KiB (very wide cout) is not an error: VALU kernel
        if (rc != SR_NOT_ELIGIBLE) return rc;
    }
    return launch_generic<bf16_t>(B, H, W, in, gates, wts, wbs, wlo, cout, ep, out, st);
}
}  // namespace sr

using namespace sr;

extern "C" int sr_conv3x3(int dtype, int B, int H, int W, const sr_concat* in, const float* gates, const void* weights,
                          long w_bstride, long w_lo_offset, int cout, const sr_epilogue* ep, sr_view out, void* stream) {
    SR_REQUIRE(dtype_ok(dtype) && B > 0 && H > 0 && W > 0 && in && weights && cout > 0 && out.p);
    SR_REQUIRE(B <= 65535 && w_bstride >= 0 && w_lo_offset >= 0);
    SR_REQUIRE(!in->ups || (H % 2 == 0 && W % 2 == 0));
    const long npix = (long)H * W;
    Concat c;
    int rc = to_concat(in, in->ups ? npix / 4 : npix, npix, &c);
    if (rc != SR_OK) return rc;
    SR_REQUIRE(!c.bands || gates);
    SR_REQUIRE(out.C == cout && out.c0 >= 0 && (out.cbs || out.c0 + cout <= out.ld) && view_block_ok(out));
    SR_REQUIRE(!ep || (view_block_ok(ep->r1) && view_block_ok(ep->r2)));
    Epilogue e = to_epilogue(ep, npix);
    SR_REQUIRE(!e.r1.p || e.r1.C == cout);
    SR_REQUIRE(!e.r2.p || e.r2.C == cout);
    int rc2 = conv3x3_launch(dtype, B, H, W, c, gates, weights, w_bstride, w_lo_offset, cout, e, to_view(out, npix), (hipStream_t)stream);
    return rc2;
}

// test hook (not part of the product API): 1 routes bf16 convs through the VALU kernel
extern "C" void sr_debug_force_generic_conv(int on) { sr::g_force_generic = on; }

extern "C" void sr_debug_split_two_pass(int on) { sr::g_split_two_pass = on; }

// test hook (A/B): 1 = never keep shared weights resident in LDS
extern "C" void sr_debug_no_resident_weights(int on) { sr::g_no_resident = on; }
