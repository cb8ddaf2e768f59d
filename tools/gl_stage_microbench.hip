// Stage-by-stage cost of the Griffin-Lim frame pipeline (uses the device code of griffin_lim.hip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/gl_stage_mb.bin tools/gl_stage_microbench.hip
#include "../single-speaker-tts_amd/csrc/griffin_lim.hip"
#include <cstdio>
#include <vector>
using namespace tts;

struct MbFrameRegs {   // the raw-row prefetch of the first kernel generation (|S| and phase rows)
    float4 m[4];
    float4 pa[4], pb[4];
    float mn, pn;
};

// PHASE 0 = B (forward), 1 = A (inverse); STAGE selects how much of the frame pipeline runs
template <int PHASE, int STAGE>
__global__ __launch_bounds__(GL_THREADS) void stage_kernel(const cf* tw1024, const cf* tw2048, const float* window,
                                                           const float* mag, const cf* phase, cf* out, float* sink,
                                                           int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int win = 1102, hop = 275, wpad = 473;
    cf* ex_all = reinterpret_cast<cf*>(smem_raw);
    cf* twR = ex_all + GL_NW * EX_CPLX;
    cf* twA = twR + 1024;
    float* wtab = reinterpret_cast<float*>(twA + 15 * 64);
    float* sig = wtab + 1104;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cf* ex = ex_all + wave * EX_CPLX;
    for (int i = tid; i < win; i += GL_THREADS) wtab[i] = window[i];
    for (int i = tid; i < 11840; i += GL_THREADS) sig[i] = 0.001f * (i % 97);
    for (int i = tid; i < 1024; i += GL_THREADS) twR[i] = tw2048[i];
    for (int i = tid; i < 15 * 64; i += GL_THREADS) twA[i] = tw1024[(i & 63) * ((i >> 6) + 1)];
    FftTw tw;
    for (int d = 1; d < 4; ++d) tw.b[d - 1] = tw1024[16 * (lane & 15) * d];
    tw.a = twA + lane;
    __syncthreads();
    float wreg[16][2];
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int nw = 2 * (lane + 64 * c) + e - wpad;
            wreg[c][e] = (nw >= 0 && nw < win) ? wtab[nw] * (0.5f / 1024.f) : 0.f;
        }
    float acc = 0.f;
    const size_t FP = 1028;
    const size_t row0 = ((size_t)blockIdx.x * GL_NW + wave) * 40;
    cf v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = cmk(0.001f * (lane + j), 0.002f * (lane - j));
    if (PHASE == 0) {
        for (int it = 0; it < iters; ++it) {
            const int fb = it % 32;
            if (STAGE >= 1) {
                const float* sf = sig + (fb + 4) * hop;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int n = 2 * (lane + 64 * j);
                    const int nw0 = n - wpad, nw1 = n + 1 - wpad;
                    float x0 = 0.f, x1 = 0.f;
                    if (nw0 >= 0 && nw0 < win) x0 = wreg[j][0] * sf[nw0];
                    if (nw1 >= 0 && nw1 < win) x1 = wreg[j][1] * sf[nw1];
                    v[j] = cmk(x0 + 1e-9f * acc, x1);
                }
            }
            fft1024(v, ex, tw, lane);
            if (STAGE == 7 || STAGE == 8 || STAGE == 9) {
                // mirror Z[M-k] through ds_bpermute (no LDS writes): lane l reg c <- lane (64-l)&63 reg 15-c,
                // lane 0 <- own reg (16-c)&15
                cf* orow = out + (row0 + (it % 40)) * FP;
                const int src = ((64 - lane) & 63) << 2;
                cf xo[16];
                if (STAGE == 9) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) ex[lane + 64 * c] = v[c];
                    wave_lds_sync();
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const int k = lane + 64 * c;
                    const cf zk = v[c];
                    if (STAGE == 9) {
                        const cf zm = cconj(ex[(MH - k) & (MH - 1)]);
                        const cf e = cadd(zk, zm);
                        const cf o = cmul(twR[k], csub(zk, zm));
                        xo[c] = unit_phasor(cadd(e, cmul_mi(o)));
                        continue;
                    }
                    float mx = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, v[15 - c].x)));
                    float my = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, v[15 - c].y)));
                    if (lane == 0) { mx = v[(16 - c) & 15].x; my = v[(16 - c) & 15].y; }
                    const cf zm = cmk(mx, -my);
                    const cf e = cadd(zk, zm);
                    const cf o = cmul(twR[k], csub(zk, zm));
                    const cf x = cadd(e, cmul_mi(o));
                    xo[c] = unit_phasor(x);
                }
                if (STAGE == 9) wave_lds_sync();
                if (STAGE == 7) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) orow[lane + 64 * c] = xo[c];
                } else {
                    // 16-byte stores: lane pairs exchange so that even lanes own (k, k+1) of row c and odd
                    // lanes (k-1, k) of row c+1
                    const bool odd = lane & 1;
#pragma unroll
                    for (int c = 0; c < 16; c += 2) {
                        const cf give = odd ? xo[c] : xo[c + 1];        // what the partner needs from me
                        cf got;
                        got.x = __shfl_xor(give.x, 1);
                        got.y = __shfl_xor(give.y, 1);
                        const cf mine = odd ? xo[c + 1] : xo[c];
                        float4 st = odd ? make_float4(got.x, got.y, mine.x, mine.y) : make_float4(mine.x, mine.y, got.x, got.y);
                        const int kk = (odd ? lane - 1 : lane) + 64 * (odd ? c + 1 : c);
                        *reinterpret_cast<float4*>(orow + kk) = st;
                    }
                }
            } else
            if (STAGE >= 2) {
#pragma unroll
                for (int c = 0; c < 16; ++c) ex[lane + 64 * c] = v[c];
                wave_lds_sync();
                cf* orow = out + (row0 + (it % 40)) * FP;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const int k = lane + 64 * c;
                    const cf zk = v[c];
                    const cf zm = cconj(ex[(MH - k) & (MH - 1)]);
                    const cf e = cadd(zk, zm);
                    const cf o = cmul(twR[k], csub(zk, zm));
                    const cf x = cadd(e, cmul_mi(o));
                    if (STAGE >= 3) orow[k] = unit_phasor(x);
                    else acc += x.x + x.y;
                }
                wave_lds_sync();
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = cscale(v[j], 1.0f / 32.0f);
            }
        }
    } else {
        MbFrameRegs nxt;
        float touch = 0.f;
        const size_t rbase = (STAGE == 4) ? 0 : row0;
        auto load_frame = [&](int f) {
            const float* mrow = mag + (rbase + f) * FP;
            const cf* prow = phase + (rbase + f) * FP;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int k = 4 * lane + 256 * jj;
                nxt.m[jj] = *reinterpret_cast<const float4*>(mrow + k);
                nxt.pa[jj] = *reinterpret_cast<const float4*>(prow + k);
                nxt.pb[jj] = *reinterpret_cast<const float4*>(prow + k + 2);
            }
            nxt.mn = 0.f; nxt.pn = 0.f;
            if (lane == 0) { nxt.mn = mrow[MH]; nxt.pn = prow[MH].x; }
        };
        if (STAGE >= 3) load_frame(0);
        else {
            for (int jj = 0; jj < 4; ++jj) { nxt.m[jj] = make_float4(1.f, .5f, .25f, 2.f); nxt.pa[jj] = make_float4(1.f, 0.f, 0.f, 1.f); nxt.pb[jj] = nxt.pa[jj]; }
            nxt.mn = 1.f; nxt.pn = 1.f;
        }
        if (STAGE == 6) {
            for (int it = 0; it < iters; ++it) {
                for (int jj = 0; jj < 4; ++jj) acc += nxt.m[jj].x + nxt.pa[jj].y + nxt.pb[jj].z;
                load_frame((it + 1) % 40);
            }
        } else
        for (int it = 0; it < iters; ++it) {
            if (STAGE >= 1) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int k = 4 * lane + 256 * jj;
                    const float4 m4 = nxt.m[jj], pa = nxt.pa[jj], pb = nxt.pb[jj];
                    float4 xa, xb;
                    xa.x = m4.x * pa.x; xa.y = m4.x * pa.y; xa.z = m4.y * pa.z; xa.w = m4.y * pa.w;
                    xb.x = m4.z * pb.x; xb.y = m4.z * pb.y; xb.z = m4.w * pb.z; xb.w = m4.w * pb.w;
                    *reinterpret_cast<float4*>(ex + k) = xa;
                    *reinterpret_cast<float4*>(ex + k + 2) = xb;
                }
                if (lane == 0) ex[MH] = cmk(nxt.mn * nxt.pn, 0.f);
                if (STAGE >= 3) load_frame((it + 1) % 40);
                if (STAGE == 5) {   // L2-warming touch two frames ahead: one dword per 128-B line
                    const int f2 = (it + 3) % 40;
                    const float ta = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(mag + (rbase + f2) * FP) + lane * 64);
                    const float tb = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(phase + (rbase + f2) * FP) + lane * 128);
                    touch += ta + tb;
                }
                wave_lds_sync();
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int k = lane + 64 * j;
                    cf xk = ex[k];
                    cf xm = cconj(ex[MH - k]);
                    if (k == 0) { xk.y = 0.f; xm.y = 0.f; }
                    const cf e = cadd(xk, xm);
                    const cf o = cmul(cconj(twR[k]), csub(xk, xm));
                    const cf zin = cadd(e, cmul_pi(o));
                    v[j] = cconj(zin);
                }
                wave_lds_sync();
            }
            fft1024(v, ex, tw, lane);
            if (STAGE >= 2) {
#pragma unroll
                for (int c = 0; c < 16; ++c) v[c] = cmk(v[c].x * wreg[c][0], -v[c].y * wreg[c][1]);
                float* sf = sig + ((it % 5) + 5 * wave) * hop;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const int n = 2 * (lane + 64 * c);
                    const int nw0 = n - wpad, nw1 = n + 1 - wpad;
                    if (nw0 >= 0 && nw0 < win) sf[nw0] += v[c].x;
                    if (nw1 >= 0 && nw1 < win) sf[nw1] += v[c].y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = cscale(v[j], 1.0f / 32.0f);
            }
        }
        acc += touch;
    }
    float s = acc;
    for (int j = 0; j < 16; ++j) s += v[j].x + v[j].y;
    sink[blockIdx.x * GL_THREADS + tid] = s + sig[tid];
}

template <int PHASE, int STAGE>
void run(const char* name, const cf* t1, const cf* t2, const float* win, const float* mag, const cf* ph, cf* out, float* sink) {
    const size_t lds = 147 * 1024;
    auto k = stage_kernel<PHASE, STAGE>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 160;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(256), dim3(GL_THREADS), lds, 0, t1, t2, win, mag, ph, out, sink, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(256), dim3(GL_THREADS), lds, 0, t1, t2, win, mag, ph, out, sink, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-44s %.3f us per frame per wave  (%s)\n", name, ms * 1e3 / iters, hipGetErrorString(hipGetLastError()));
}

int main() {
    std::vector<cf> t1(1024), t2(1024);
    for (int k = 0; k < 1024; ++k) {
        const double a1 = -2.0 * M_PI * k / 1024.0, a2 = -2.0 * M_PI * k / 2048.0;
        t1[k] = make_float2((float)cos(a1), (float)sin(a1));
        t2[k] = make_float2((float)cos(a2), (float)sin(a2));
    }
    std::vector<float> w(1102);
    for (int i = 0; i < 1102; ++i) w[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / 1102));
    cf *d1, *d2, *ph, *out; float *dw, *mag, *sink;
    const size_t rows = (size_t)256 * 8 * 40, FP = 1028;
    hipMalloc(&d1, 8192); hipMalloc(&d2, 8192); hipMalloc(&dw, 1102 * 4);
    hipMalloc(&mag, rows * FP * 4); hipMalloc(&ph, rows * FP * 8); hipMalloc(&out, rows * FP * 8); hipMalloc(&sink, 256 * 512 * 4);
    hipMemset(mag, 0, rows * FP * 4); hipMemset(ph, 0, rows * FP * 8);
    hipMemcpy(d1, t1.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(d2, t2.data(), 8192, hipMemcpyHostToDevice);
    hipMemcpy(dw, w.data(), 1102 * 4, hipMemcpyHostToDevice);
    run<0, 0>("B0: FFT only", d1, d2, dw, mag, ph, out, sink);
    run<0, 1>("B1: + windowed loads from LDS signal", d1, d2, dw, mag, ph, out, sink);
    run<0, 2>("B2: + mirror exchange + split pass", d1, d2, dw, mag, ph, out, sink);
    run<0, 3>("B3: + unit phasor + global stores", d1, d2, dw, mag, ph, out, sink);
    run<0, 7>("B7: bpermute mirror + dwordx2 stores", d1, d2, dw, mag, ph, out, sink);
    run<0, 8>("B8: bpermute mirror + dwordx4 stores", d1, d2, dw, mag, ph, out, sink);
    run<0, 9>("B9: LDS mirror + dwordx4 stores", d1, d2, dw, mag, ph, out, sink);
    run<1, 0>("A0: FFT only", d1, d2, dw, mag, ph, out, sink);
    run<1, 1>("A1: + X exchange + merge pass", d1, d2, dw, mag, ph, out, sink);
    run<1, 2>("A2: + window scale + overlap-add RMW", d1, d2, dw, mag, ph, out, sink);
    run<1, 3>("A3: + prefetched global loads (HBM)", d1, d2, dw, mag, ph, out, sink);
    run<1, 4>("A4: same, rows L2-resident", d1, d2, dw, mag, ph, out, sink);
    run<1, 5>("A5: HBM + L2 touch 3 frames ahead", d1, d2, dw, mag, ph, out, sink);
    run<1, 6>("A6: loads only (HBM), 1 frame in flight/wave", d1, d2, dw, mag, ph, out, sink);
    return 0;
}
