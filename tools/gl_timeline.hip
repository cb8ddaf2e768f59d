// Timeline of one Griffin-Lim workgroup (diagnostic build of the production kernel with -DGL_STAMPS).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGL_STAMPS -o tools/gl_timeline.bin tools/gl_timeline.hip
#include "../single-speaker-tts_amd/csrc/griffin_lim.hip"
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>
using namespace tts;

int main() {
    const int B = 64, T = 1000, FP = 1028, win = 1102, hop = 275;
    std::vector<cf> t1(1024), t2(1024);
    for (int k = 0; k < 1024; ++k) {
        const double a1 = -2.0 * M_PI * k / 1024.0, a2 = -2.0 * M_PI * k / 2048.0;
        t1[k] = make_float2((float)cos(a1), (float)sin(a1));
        t2[k] = make_float2((float)cos(a2), (float)sin(a2));
    }
    std::vector<float> w(win), wss(2048 + hop * (T - 1), 1.0f);
    for (int i = 0; i < win; ++i) w[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / win));
    cf *d1, *d2, *xin, *xout; float *dw, *dwss, *mag; unsigned long long* dbg;
    const size_t n = (size_t)B * T * FP;
    hipMalloc(&d1, 8192); hipMalloc(&d2, 8192); hipMalloc(&dw, win * 4); hipMalloc(&dwss, wss.size() * 4);
    hipMalloc(&mag, n * 4); hipMalloc(&xin, n * 8); hipMalloc(&xout, n * 8); hipMalloc(&dbg, 1 << 20);
    hipMemset(dbg, 0, 1 << 20);
    {   // non-trivial data (all-zero input would send every frame down the exact-phasor path)
        std::vector<float> hm(n);
        std::vector<cf> hx(n);
        unsigned st = 12345u;
        for (size_t i = 0; i < n; ++i) {
            st = st * 1664525u + 1013904223u;
            const float m = 0.01f + (float)(st >> 8) * (1.0f / 16777216.0f);
            st = st * 1664525u + 1013904223u;
            const float a = 6.2831853f * (float)(st >> 8) * (1.0f / 16777216.0f);
            hm[i] = m;
            hx[i] = make_float2(m * cosf(a), m * sinf(a));
        }
        hipMemcpy(mag, hm.data(), n * 4, hipMemcpyHostToDevice);
        hipMemcpy(xin, hx.data(), n * 8, hipMemcpyHostToDevice);
    }
    hipMemcpy(d1, t1.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(d2, t2.data(), 8192, hipMemcpyHostToDevice);
    hipMemcpy(dw, w.data(), win * 4, hipMemcpyHostToDevice);
    hipMemcpy(dwss, wss.data(), wss.size() * 4, hipMemcpyHostToDevice);
    GlParams p;
    memset(&p, 0, sizeof(p));
    p.mag = mag; p.phase_in = xin; p.phase_out = xout; p.wav = reinterpret_cast<float*>(dbg);
    cf* dtab; hipMalloc(&dtab, (1024 + 960) * 8);
    { std::vector<cf> tb(1984); for (int k = 0; k < 1024; ++k) tb[k] = t2[k]; for (int i = 0; i < 960; ++i) tb[1024 + i] = t1[(i & 63) * ((i >> 6) + 1)]; hipMemcpy(dtab, tb.data(), 1984 * 8, hipMemcpyHostToDevice); }
    p.window = dw; p.rwss = dwss; p.tw1024 = d1; p.tw2048 = d2; p.tables = dtab;
    p.T = T; p.FP = FP; p.win = win; p.hop = hop; p.B = B; p.C = 32; p.ncol = 5;
    gl_configure();
    for (int it = 0; it < 3; ++it) launch_gl_iter(0, p, B, 0);
    hipDeviceSynchronize();
    printf("launch: %s\n", hipGetErrorString(hipGetLastError()));
    const int nwg = (2048 + 96) / 97;
    std::vector<unsigned long long> h((size_t)nwg * GL_NW * 16);
    hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
#ifdef GL_STAMPS_FINE
    const char* names[13] = {"start", "A2:begin", "A2:split+loads", "A2:fft+win", "A2:flag", "A2:ola", "B1:begin", "B1:loaded", "B1:fft", "B1:merge+store", "-", "-", "-"};
#else
    const char* names[13] = {"start", "prologue", "A0", "A1", "A2", "A3", "A4", "A-barrier", "normalised", "B0", "B1", "B2", "B3"};
#endif
    // average over sampled WGs, per wave: time since WG's earliest start
    for (int wv = 0; wv < GL_NW; ++wv) {
        printf("wave %d:", wv);
        for (int s = 0; s < 13; ++s) {
            double acc = 0; int cnt = 0;
            for (int g = 0; g < nwg - 1; ++g) {
                unsigned long long t0 = ~0ull;
                for (int w2 = 0; w2 < GL_NW; ++w2) if (h[((size_t)g * GL_NW + w2) * 16]) t0 = std::min(t0, h[((size_t)g * GL_NW + w2) * 16]);
                const unsigned long long v = h[((size_t)g * GL_NW + wv) * 16 + s];
                if (v && t0 != ~0ull) { acc += (double)(v - t0) * 0.01; ++cnt; }   // 100 MHz -> us
            }
            printf(" %s=%.1f", names[s], cnt ? acc / cnt : -1.0);
        }
        printf("\n");
    }
    return 0;
}
