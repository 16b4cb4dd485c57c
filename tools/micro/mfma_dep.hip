// Does a chain of DEPENDENT v_mfma_f32_32x32x2_f32 (the accumulation chain of one distance tile, csrc/vq.hip) issue at full rate?
// One wave per SIMD (256 threads per block, one block per CU), 4096 MFMAs per wave: one chain, or two / four interleaved chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(256) void chains(float* out, float a0, int iters) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64 / CH; ++k)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        a += 1e-6f;
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH> void run(float* out, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 256;
    hipLaunchKernelGGL(chains<CH>, dim3(256), dim3(256), 0, 0, out, 0.5f, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(chains<CH>, dim3(256), dim3(256), 0, 0, out, 0.5f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = 256.0 * 4 * iters * 64 * 4096.0;
    printf("%s: %.3f ms, %.1f TF (peak 157)\n", name, ms, flops / ms * 1e-9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    run<1>(out, "1 chain ");
    run<2>(out, "2 chains");
    run<4>(out, "4 chains");
    return 0;
}
