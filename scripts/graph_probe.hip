// Does replaying a captured hipGraph of short dependent kernels beat launching them on a stream?  Two ~4 us kernels per "iteration" (the shape of the 2D
// visco-elastic loop at 512^2), 100 iterations per graph.   hipcc --offload-arch=gfx950 -O3 scripts/graph_probe.hip -o gpurun_out/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_a(double *x, const double *y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = 0.5 * x[i] + 0.25 * (y[i] + y[(i + 1) % n]); }
int main()
{
    const int n = 513 * 513, iters = 100, reps = 40;
    double *x, *y;
    hipMalloc(&x, n * 8); hipMalloc(&y, n * 8); hipMemset(x, 0, n * 8); hipMemset(y, 0, n * 8);
    hipStream_t s; hipStreamCreate(&s);
    const dim3 g((n + 255) / 256), b(256);
    auto body = [&]() { for (int it = 0; it < iters; it++) { hipLaunchKernelGGL(k_a, g, b, 0, s, x, y, n); hipLaunchKernelGGL(k_a, g, b, 0, s, y, x, n); } };
    body(); hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) body();
    hipStreamSynchronize(s);
    double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * iters);
    hipGraph_t graph; hipGraphExec_t exec;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal); body(); hipStreamEndCapture(s, &graph);
    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphLaunch(exec, s); hipStreamSynchronize(s);
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) hipGraphLaunch(exec, s);
    hipStreamSynchronize(s);
    double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * iters);
    printf("per iteration (2 kernels): stream launches %.2f us, graph replay %.2f us\n", us_stream, us_graph);
    return 0;
}
