// Device helpers shared by the decode kernels (decode_kernels.hip, decode_peaks.hip).  Every translation unit that includes
// this is built with -ffp-contract=off: each fused multiply-add below is explicit because the results must be bit-identical to
// the reference's torch-CPU / numpy arithmetic (oracle/decode_oracle.c holds the experimentally pinned formulas).
#pragma once
#include "decode_kernels.h"

#include <math.h>

typedef unsigned long long u64;

// element idx of a wave-uniform plane: the byte offset stays 32-bit (planes are < 2^24 pixels), so the load takes the scalar-base +
// vector-offset form and needs one address register, not a 64-bit pair
__device__ __forceinline__ float ldg(const float *__restrict__ base, int idx)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (unsigned)(idx << 2));
}

// ------------------------------------------------------------------ bilinear sampling
// F.interpolate(mode="bilinear", align_corners=False), torch CPU fp32 (results.py:48-67)
struct Lin { int i0, i1; float w0, w1; };

__device__ __forceinline__ Lin src_index(int in_size, float scale, int dst)
{
    float r = __builtin_fmaf(scale, (float)dst + 0.5f, -0.5f);
    if (r < 0.f) r = 0.f;
    const int a = (int)r;
    float l1 = r - (float)a;
    l1 = fminf(fmaxf(l1, 0.f), 1.f);
    Lin o;
    o.i0 = a; o.i1 = a + (a < in_size - 1 ? 1 : 0); o.w1 = l1; o.w0 = 1.f - l1;
    return o;
}

__device__ __forceinline__ float bilerp(const float *__restrict__ img, int w, const Lin &ly, const Lin &lx)
{
    const float *r0 = img + (size_t)ly.i0 * w, *r1 = img + (size_t)ly.i1 * w;
    const float a = __builtin_fmaf(r0[lx.i0], lx.w0, r0[lx.i1] * lx.w1);
    const float b = __builtin_fmaf(r1[lx.i0], lx.w0, r1[lx.i1] * lx.w1);
    return __builtin_fmaf(a, ly.w0, b * ly.w1);
}

// one value of the stage average (results.py:225-226: the 1/4-res heatmaps resized to 1/2 res, mean of the two stages)
__device__ __forceinline__ float avg_at(const DecodeSrc &s, int b, int k, int r, int c)
{
    const int hq = s.H >> 2, wq = s.W >> 2, wh = s.W >> 1;
    const float *q = s.hm_q + (size_t)b * s.hm_q_bs + (size_t)k * hq * wq;
    const float up = bilerp(q, wq, src_index(hq, 0.5f, r), src_index(wq, 0.5f, c));
    return (up + s.hm_h[(size_t)b * s.hm_h_bs + ((size_t)k * (s.H >> 1) + r) * wh + c]) / 2.0f;
}

// full-resolution heat value / tag value at (b,k,y,x)
__device__ __forceinline__ float heat_at(const DecodeSrc &s, int b, int k, int y, int x)
{
    if (s.mode == 1) return s.hm_full[(((size_t)b * s.K + k) * s.H + y) * s.W + x];
    const int hh = s.H >> 1, wh = s.W >> 1;
    if (s.avg) return bilerp(s.avg + ((size_t)b * s.K + k) * hh * wh, wh, src_index(hh, s.scale_h2, y), src_index(wh, s.scale_w2, x));
    // no materialised stage average: the four half-res taps are formed here, each as stage_average_kernel would have stored it
    const Lin ly = src_index(hh, s.scale_h2, y), lx = src_index(wh, s.scale_w2, x);
    const float a = __builtin_fmaf(avg_at(s, b, k, ly.i0, lx.i0), lx.w0, avg_at(s, b, k, ly.i0, lx.i1) * lx.w1);
    const float c = __builtin_fmaf(avg_at(s, b, k, ly.i1, lx.i0), lx.w0, avg_at(s, b, k, ly.i1, lx.i1) * lx.w1);
    return __builtin_fmaf(a, ly.w0, c * ly.w1);
}
// N full-resolution heat values of map (b, k) at once on the default path (mode 0, no materialised stage average): every sample is
// 4 averaged taps x (4 quarter-res + 1 half-res) loads, and asked for one after the other through heat_at() (whose mode switches are
// run-time branches) each of them is a memory round trip of its own; here the 20 N loads are all issued before the first use.
// Same expressions as heat_at() / avg_at() per value.
template <int N>
__device__ __forceinline__ void heat_otf(const DecodeSrc &s, int b, int k, const int (&ys)[N], const int (&xs)[N], float (&out)[N])
{
    const int hq = s.H >> 2, wq = s.W >> 2, hh = s.H >> 1, wh = s.W >> 1;
    const float *q = s.hm_q + (size_t)b * s.hm_q_bs + (size_t)k * hq * wq;
    const float *h = s.hm_h + (size_t)b * s.hm_h_bs + (size_t)k * hh * wh;
    Lin ly[N], lx[N], qy[N][2], qx[N][2];
    float qv[N][2][2][4], hv[N][2][2];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        ly[n] = src_index(hh, s.scale_h2, ys[n]); lx[n] = src_index(wh, s.scale_w2, xs[n]);
        qy[n][0] = src_index(hq, 0.5f, ly[n].i0); qy[n][1] = src_index(hq, 0.5f, ly[n].i1);
        qx[n][0] = src_index(wq, 0.5f, lx[n].i0); qx[n][1] = src_index(wq, 0.5f, lx[n].i1);
    }
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const Lin &ry = qy[n][a], &rx = qx[n][c];
                qv[n][a][c][0] = ldg(q, ry.i0 * wq + rx.i0); qv[n][a][c][1] = ldg(q, ry.i0 * wq + rx.i1);
                qv[n][a][c][2] = ldg(q, ry.i1 * wq + rx.i0); qv[n][a][c][3] = ldg(q, ry.i1 * wq + rx.i1);
                hv[n][a][c] = ldg(h, (a ? ly[n].i1 : ly[n].i0) * wh + (c ? lx[n].i1 : lx[n].i0));
            }
#pragma unroll
    for (int n = 0; n < N; ++n) {
        float av[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const Lin &ry = qy[n][a], &rx = qx[n][c];
                const float u0 = __builtin_fmaf(qv[n][a][c][0], rx.w0, qv[n][a][c][1] * rx.w1);
                const float u1 = __builtin_fmaf(qv[n][a][c][2], rx.w0, qv[n][a][c][3] * rx.w1);
                av[a][c] = (__builtin_fmaf(u0, ry.w0, u1 * ry.w1) + hv[n][a][c]) / 2.0f;
            }
        const float t0 = __builtin_fmaf(av[0][0], lx[n].w0, av[0][1] * lx[n].w1);
        const float t1 = __builtin_fmaf(av[1][0], lx[n].w0, av[1][1] * lx[n].w1);
        out[n] = __builtin_fmaf(t0, ly[n].w0, t1 * ly[n].w1);
    }
}

__device__ __forceinline__ float tag_at(const DecodeSrc &s, int b, int k, int y, int x, int e)
{
    if (s.mode == 1) return s.tags_full[((((size_t)b * s.K + k) * s.H + y) * s.W + x) * s.E + e];
    const int hq = s.H >> 2, wq = s.W >> 2;
    return bilerp(s.tags_q[e] + (size_t)b * s.tags_bs[e] + (size_t)k * hq * wq, wq, src_index(hq, s.scale_h4, y), src_index(wq, s.scale_w4, x));
}

// ------------------------------------------------------------------ sortable keys
// Larger key = larger value; between equal values the smaller flat index wins (torch.topk
// leaves that order unspecified; the oracle uses the same rule). -0 == +0; NaN ranks lowest.
__device__ __forceinline__ u64 make_key(float v, unsigned idx)
{
    if (v != v) return 1ull + (u64)(0xffffffffu - idx);
    if (v == 0.f) v = 0.f;
    unsigned bits = __float_as_uint(v);
    bits = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
    return ((u64)bits << 32) | (u64)(0xffffffffu - idx);
}
__device__ __forceinline__ u64 wave_max_u64(u64 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u64 o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    return v;
}

// bf16 roundings toward -inf / +inf (finite inputs): truncation moves toward zero, so step away from zero when bits were lost
__device__ __forceinline__ unsigned short bf16_floor(float f)
{
    const unsigned u = __float_as_uint(f);
    unsigned short t = (unsigned short)(u >> 16);
    if ((u & 0xffffu) && (u >> 31)) ++t;  // negative and inexact: one step more negative
    return t;
}
__device__ __forceinline__ unsigned short bf16_ceil(float f)
{
    const unsigned u = __float_as_uint(f);
    unsigned short t = (unsigned short)(u >> 16);
    if ((u & 0xffffu) && !(u >> 31)) ++t;  // positive and inexact: one step more positive
    return t;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it parks the wave until every
// global store it has issued (cell maxima, candidate lists) is acknowledged by memory: a full round trip per barrier that no
// thread of the workgroup depends on.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
