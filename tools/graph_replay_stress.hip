// Stand-alone hipGraph replay stress (no library) -- what the wrong decoder-graph replays of DESIGN.md section 8 are NOT:
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/graph_replay_stress.bin tools/graph_replay_stress.hip
//   tools/bin/graph_replay_stress.bin [rounds] [nodes]
//   LD_PRELOAD=<torch>/lib/libamdhip64.so LD_LIBRARY_PATH=<torch>/lib tools/bin/graph_replay_stress.bin ...   (PyTorch's bundled runtime)
// Every round captures up to `nodes` one-workgroup kernels with a private segment (dec_gemm_kernel has 16 bytes of scratch),
// each writing (round, node) into its own 64 floats of this round's buffer (two buffers alternate and are never freed),
// instantiates the graph, replays it, runs a big resident kernel on the stream, lets the queue go idle, replays the SAME
// executable graph again (on another stream every other round), checks every value on the host and destroys the graph; between
// rounds eager kernels on a second stream and hipMalloc / hipFree of other sizes.  Result on MI355X: 0 wrong rounds of 200-1000
// on HIP 7.2.26015 (/opt/rocm) AND on 7.0.51831 (PyTorch 2.10's bundle) -- stale kernel arguments, scratch under replay and
// graph memory re-use alone do not show the problem; the library's own sequence does, on the 7.0 runtime only
// (tools/graph_probe.py --torch: 5 of 5; profiles/r06_experiment_hipgraph.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

// (a kernel with a private segment, like dec_gemm_kernel's 16 bytes of spilled registers: the dynamically indexed array lives in
// scratch memory, and what it holds decides the stamp)
__global__ void stamp(float* out, float round, float node, const float* in, int n_in) {
    volatile float priv[24];
    for (int i = 0; i < 24; ++i) priv[i] = (i & 1) ? node : round;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_in; i += 64) acc += in[i] + priv[(i + (int)in[0]) % 24] * 0.f;   // (a read as well, like a layer of the decoder)
    const int sel = (int)in[1];   // 0 at run time: the compiler cannot fold the private array away
    out[threadIdx.x] = threadIdx.x == 0 ? priv[2 * sel] : (threadIdx.x == 1 ? priv[2 * sel + 1] : acc);
}
// a big resident kernel between two replays (the persistent decoder: 512 threads, ~240 registers, a lot of LDS, no scratch)
__global__ __launch_bounds__(512) void resident(float* p, int iters) {
    extern __shared__ float lds[];
    float r[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) r[i] = p[(threadIdx.x + i * 512) & 0xFFFF];
    for (int k = 0; k < iters; ++k) {
#pragma unroll
        for (int i = 0; i < 96; ++i) r[i] = r[i] * 1.0001f + r[(i + 7) % 96];
        lds[threadIdx.x] = r[k % 96];
        __syncthreads();
    }
    float s = lds[(threadIdx.x + 1) & 511];
#pragma unroll
    for (int i = 0; i < 96; ++i) s += r[i];
    p[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ void busy(float* p, int n, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = p[i];
    for (int k = 0; k < iters; ++k) x = x * 1.0001f + 0.5f;
    p[i] = x;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200, max_nodes = argc > 2 ? atoi(argv[2]) : 2000;
    int rtv = 0;
    CHECK(hipRuntimeGetVersion(&rtv));
    hipStream_t s, s2;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    float *buf[2], *in, *work;
    const size_t buf_floats = (size_t)max_nodes * 64;
    CHECK(hipMalloc(&buf[0], buf_floats * sizeof(float)));
    CHECK(hipMalloc(&buf[1], buf_floats * sizeof(float)));
    CHECK(hipMalloc(&in, 4096 * sizeof(float)));
    CHECK(hipMalloc(&work, (1 << 20) * sizeof(float)));
    CHECK(hipMemset(in, 0, 4096 * sizeof(float)));
    CHECK(hipMemset(work, 0, (1 << 20) * sizeof(float)));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&resident), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    std::vector<float> host(buf_floats);
    int bad_rounds = 0, first_bad = -1;
    long long stale = 0, unwritten = 0;
    srand(1);
    for (int r = 0; r < rounds; ++r) {
        const int nodes = 40 + rand() % (max_nodes - 39);   // graphs of changing size share the runtime's argument memory
        float* b = buf[r & 1];
        CHECK(hipMemsetAsync(b, 0xFF, buf_floats * sizeof(float), s));   // NaN pattern: "not written"
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int n = 0; n < nodes; ++n) hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, b + (size_t)n * 64, (float)r, (float)n, in, 256);
        CHECK(hipStreamEndCapture(s, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        // replay, a big resident kernel on the same stream, an idle moment, the same executable graph AGAIN (a cached decoder
        // graph after a call of the persistent decoder), on the other stream every other round
        CHECK(hipGraphLaunch(ge, s));
        hipLaunchKernelGGL(resident, dim3(32), dim3(512), 96 * 1024, s, work, 200);
        CHECK(hipStreamSynchronize(s));
        CHECK(hipMemsetAsync(b, 0xFF, buf_floats * sizeof(float), s));
        CHECK(hipStreamSynchronize(s));
        hipStream_t ls = (r & 2) ? s2 : s;
        CHECK(hipGraphLaunch(ge, ls));
        CHECK(hipMemcpyAsync(host.data(), b, (size_t)nodes * 64 * sizeof(float), hipMemcpyDeviceToHost, ls));
        CHECK(hipStreamSynchronize(ls));
        int bad = 0;
        for (int n = 0; n < nodes; ++n) {
            const float rr = host[(size_t)n * 64], nn = host[(size_t)n * 64 + 1];
            if (rr == (float)r && nn == (float)n) continue;
            ++bad;
            if (rr != rr) ++unwritten; else ++stale;
            if (first_bad < 0) { first_bad = r; printf("round %d (%d nodes): node %d holds (round %g, node %g)\n", r, nodes, n, rr, nn); }
        }
        bad_rounds += bad != 0;
        CHECK(hipGraphExecDestroy(ge));
        CHECK(hipGraphDestroy(g));
        // what else a long-lived process does between two graphs
        hipLaunchKernelGGL(busy, dim3(4096), dim3(256), 0, s2, work, 1 << 20, 50 + rand() % 200);
        void* tmp = nullptr;
        CHECK(hipMalloc(&tmp, (size_t)(1 + rand() % 64) << 20));
        CHECK(hipFree(tmp));
    }
    CHECK(hipDeviceSynchronize());
    printf("HIP runtime %d: %d of %d rounds wrong (%lld nodes ran with another graph's arguments, %lld buffers not written); first wrong round %d\n",
           rtv, bad_rounds, rounds, stale, unwritten, first_bad);
    return bad_rounds ? 1 : 0;
}
