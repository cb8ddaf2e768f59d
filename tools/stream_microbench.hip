// Memory-side ceiling of the Griffin-Lim access pattern: persistent workgroups, one wave per spectrum row, per row
// 8 KB complex in (forward order + mirrored order, the second pass hits in cache), 4 KB magnitudes in, 8 KB complex
// out -- no FFT, a few dozen VALU per row.  Variants: bytes per lane and access (8 = what gl_iter_kernel does,
// 16 = two bins per lane), waves per CU, prefetch depth one row.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/stream_mb.bin tools/stream_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float2 cf;
#define FP 1028
#define MH 1024

template <int NW, int WIDE, int VALU, int MIRROR = 1, int NTST = 1, int ILV = 0>
__global__ __launch_bounds__(NW * 64) void stream_kernel(const cf* __restrict__ in, const float* __restrict__ mag,
                                                         cf* __restrict__ out, int rows, int rows_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * NW + wave;
    // ILV = 0: every wave walks its own block of consecutive rows; ILV = 1: a workgroup owns one block and deals its
    // rows round-robin to its waves (what gl_iter_kernel does with the frames of a run)
    const int r0 = ILV ? blockIdx.x * NW * rows_per_wave + wave : gw * rows_per_wave;
    const int rstep = ILV ? NW : 1;
    if (WIDE == 0) {
        cf gk[16], gm[16];
        float mg[16];
        auto load = [&](int r) {
            const cf* row = in + (size_t)r * FP;
#pragma unroll
            for (int j = 0; j < 16; ++j) gk[j] = row[lane + 64 * j];
#pragma unroll
            for (int j = 0; j < 16; ++j) gm[j] = MIRROR ? row[MH - lane - 64 * j] : gk[15 - j];
        };
        if (r0 < rows) load(r0);
        for (int i = 0; i < rows_per_wave; ++i) {
            const int r = r0 + i * rstep;
            if (r >= rows) break;
            cf v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = make_float2(gk[j].x + gm[j].x, gk[j].y - gm[j].y);
            const float* mrow = mag + (size_t)r * FP;
#pragma unroll
            for (int j = 0; j < 16; ++j) mg[j] = __builtin_nontemporal_load(mrow + lane + 64 * j);
            if (i + 1 < rows_per_wave && r + rstep < rows) load(r + rstep);
#pragma unroll
            for (int q = 0; q < VALU; ++q)
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = make_float2(fmaf(v[j].x, 1.0001f, v[j].y), fmaf(v[j].y, 0.9999f, -v[j].x));
            cf* orow = out + (size_t)r * FP;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 t;
                t.x = v[j].x * mg[j];
                t.y = v[j].y * mg[j];
                if (NTST) __builtin_nontemporal_store(t, reinterpret_cast<f2*>(orow + lane + 64 * j));
                else *reinterpret_cast<f2*>(orow + lane + 64 * j) = t;
            }
        }
    } else {
        // two consecutive bins per lane: even lanes take the even 64-bin groups, odd lanes the odd ones
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const int pe = lane & ~1, par = lane & 1;
        f4 gk[8], gm[8];
        auto load = [&](int r) {
            const float* row = reinterpret_cast<const float*>(in + (size_t)r * FP);
#pragma unroll
            for (int s = 0; s < 8; ++s) gk[s] = *reinterpret_cast<const f4*>(row + 2 * (pe + 64 * (2 * s + par)));
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                // bins MH - k - 1, MH - k for k = pe + 64 (2 s + par): 8-byte aligned 16-byte load
                const float* p = row + 2 * (MH - (pe + 64 * (2 * s + par)) - 1);
                f2 a = *reinterpret_cast<const f2*>(p), b = *reinterpret_cast<const f2*>(p + 2);
                gm[s] = (f4){a.x, a.y, b.x, b.y};
            }
        };
        if (r0 < rows) load(r0);
        for (int i = 0; i < rows_per_wave; ++i) {
            const int r = r0 + i;
            if (r >= rows) break;
            f4 v[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) v[s] = gk[s] + gm[s];
            const float* mrow = mag + (size_t)r * FP;
            f2 mg[8];
#pragma unroll
            for (int s = 0; s < 8; ++s)
                mg[s] = __builtin_nontemporal_load(reinterpret_cast<const f2*>(mrow + pe + 64 * (2 * s + par)));
            if (i + 1 < rows_per_wave && r + 1 < rows) load(r + 1);
#pragma unroll
            for (int q = 0; q < VALU; ++q)
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    v[s].x = fmaf(v[s].x, 1.0001f, v[s].y); v[s].y = fmaf(v[s].y, 0.9999f, -v[s].x);
                    v[s].z = fmaf(v[s].z, 1.0001f, v[s].w); v[s].w = fmaf(v[s].w, 0.9999f, -v[s].z);
                }
            float* orow = reinterpret_cast<float*>(out + (size_t)r * FP);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                f4 t = v[s];
                t.x *= mg[s].x; t.y *= mg[s].x; t.z *= mg[s].y; t.w *= mg[s].y;
                __builtin_nontemporal_store(t, reinterpret_cast<f4*>(orow + 2 * (pe + 64 * (2 * s + par))));
            }
        }
    }
}

// gl_iter_kernel's phase structure: all waves of a workgroup first only READ spectrum rows (phase A: CH rows each, kept
// as a checksum), barrier, then only read magnitudes and WRITE rows (phase B), barrier.  SPLIT = 1: the two halves of the
// workgroup run the phases in opposite order (one half reads while the other writes).
template <int NW, int CH, int SPLIT>
__global__ __launch_bounds__(NW * 64) void phased_kernel(const cf* __restrict__ in, const float* __restrict__ mag, cf* __restrict__ out, int rows,
                                                         int rows_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = blockIdx.x * rows_per_wg;
    const int rounds = rows_per_wg / (NW * CH);
    const bool flip = SPLIT && wave >= NW / 2;
    float acc = 0.f;
    for (int rd = 0; rd < 2 * rounds + (SPLIT ? 1 : 0); ++rd) {
        const int ph = (rd + (flip ? 1 : 0)) & 1;           // 0 = read phase, 1 = write phase
        const int blk = (rd - (flip ? 1 : 0)) >> 1;         // which block of rows
        if (blk >= 0 && blk < rounds && (rd - (flip ? 1 : 0)) >= 0) {
            for (int c = 0; c < CH; ++c) {
                const int r = base + (blk * CH + c) * NW + wave;
                if (r >= rows) break;
                if (ph == 0) {
                    const cf* row = in + (size_t)r * FP;
                    cf g[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) g[j] = row[lane + 64 * j];
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc += g[j].x + g[j].y;
                } else {
                    const float* mrow = mag + (size_t)r * FP;
                    cf* orow = out + (size_t)r * FP;
                    float mg[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) mg[j] = __builtin_nontemporal_load(mrow + lane + 64 * j);
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        f2 t; t.x = mg[j] + acc; t.y = mg[j] - acc;
                        __builtin_nontemporal_store(t, reinterpret_cast<f2*>(orow + lane + 64 * j));
                    }
                }
            }
        }
        __syncthreads();
    }
}
template <int NW, int CH, int SPLIT>
void run_phased(const cf* in, const float* mag, cf* out, int rows, int wgs) {
    const int per = NW * CH;
    const int rpwg = ((rows + wgs - 1) / wgs + per - 1) / per * per;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((phased_kernel<NW, CH, SPLIT>), dim3(wgs), dim3(NW * 64), 0, 0, in, mag, out, rows, rpwg);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep) best = ms < best ? ms : best;
    }
    printf("phased NW=%d rows per wave and phase=%d split=%d wgs=%d: %.1f us, %.2f TB/s algorithmic (%s)\n", NW, CH, SPLIT, wgs, best * 1e3,
           20.0 * 1025 * rows / (best * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

template <int NW, int WIDE, int VALU, int MIRROR = 1, int NTST = 1, int ILV = 0>
void run(const cf* in, const float* mag, cf* out, int rows, int wgs) {
    const int waves = wgs * NW;
    const int rpw = (rows + waves - 1) / waves;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((stream_kernel<NW, WIDE, VALU, MIRROR, NTST, ILV>), dim3(wgs), dim3(NW * 64), 0, 0, in, mag, out, rows, rpw);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((stream_kernel<NW, WIDE, VALU, MIRROR, NTST, ILV>), dim3(wgs), dim3(NW * 64), 0, 0, in, mag, out, rows, rpw);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    const double bytes = 20.0 * 1025 * rows;
    printf("NW=%2d wide=%d valu=%3d mirror=%d nt=%d ilv=%d wgs=%d: %.1f us, %.2f TB/s algorithmic (%s)\n", NW, WIDE, VALU * 32, MIRROR, NTST, ILV, wgs, best * 1e3,
           bytes / (best * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int rows = 64000;
    cf *in, *out;
    float* mag;
    (void)hipMalloc(&in, (size_t)rows * FP * sizeof(cf));
    (void)hipMalloc(&out, (size_t)rows * FP * sizeof(cf));
    (void)hipMalloc(&mag, (size_t)rows * FP * sizeof(float));
    (void)hipMemset(in, 0x11, (size_t)rows * FP * sizeof(cf));
    (void)hipMemset(mag, 0x22, (size_t)rows * FP * sizeof(float));
    run<8, 0, 0, 0, 1, 0>(in, mag, out, rows, 256);
    run<8, 0, 0, 0, 1, 1>(in, mag, out, rows, 256);
    run<8, 0, 0, 0, 1, 1>(in, mag, out, rows, 224);
    run<8, 1, 0, 0, 1, 0>(in, mag, out, rows, 256);
    run<4, 0, 0, 0, 1, 1>(in, mag, out, rows, 256);
    run<4, 0, 0, 0, 1, 1>(in, mag, out, rows, 512);
    run<12, 0, 0, 0, 1, 1>(in, mag, out, rows, 256);
    run<16, 0, 0, 0, 1, 1>(in, mag, out, rows, 256);
    run<8, 0, 40, 0, 1, 1>(in, mag, out, rows, 256);
    run<8, 0, 20, 0, 1, 1>(in, mag, out, rows, 256);
    run<12, 0, 20, 0, 1, 1>(in, mag, out, rows, 256);
    run_phased<8, 8, 0>(in, mag, out, rows, 256);
    run_phased<8, 8, 1>(in, mag, out, rows, 256);
    run_phased<8, 4, 0>(in, mag, out, rows, 256);
    run_phased<8, 4, 1>(in, mag, out, rows, 256);
    run_phased<8, 1, 0>(in, mag, out, rows, 256);
    run_phased<8, 1, 1>(in, mag, out, rows, 256);
    return 0;
}
