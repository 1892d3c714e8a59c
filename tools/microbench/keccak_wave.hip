// SHAKE-256 with one Keccak state per WAVE (fusion-cryptography_amd/csrc/fz_keccak_wave.h): correctness against vectors made by
// hashlib (tools/probes/keccak_wave_check.py writes the inputs and compares the outputs) and the time of the chain the
// challenge pipeline runs per signer (47 absorbed + 61 squeezed blocks).
//   keccak_wave <in.bin> <out.bin> <N> <stride> <out_blocks> [reps]
// in.bin: N rows of `stride` bytes (padded SHAKE-256 blocks) followed by N int32 block counts; out.bin: N * out_blocks * 136 bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../fusion-cryptography_amd/csrc/fz_keccak_wave.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int W>
__global__ __launch_bounds__(64 * W) void shake_wave(const uint8_t *text, size_t stride, const int *nblocks, size_t N, int out_blocks, uint8_t *out) {
    const int lane = threadIdx.x & 63;
    const size_t s = (size_t)blockIdx.x * W + (threadIdx.x >> 6);
    if (s >= N) return;
    fzkw::Wave K;
    K.init(lane);
    const int nb = nblocks[s];
    const bool ab = K.word < 17;
    const uint2 *row = reinterpret_cast<const uint2 *>(text + s * stride) + (ab ? K.word : 0);
    uint2 m = ab ? row[0] : make_uint2(0u, 0u);
#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
        K.lo ^= m.x;
        K.hi ^= m.y;
        if (ab) m = row[(size_t)(b + 1 < nb ? b + 1 : b) * 17];
        K.permute();
    }
    uint2 *o = reinterpret_cast<uint2 *>(out + s * (size_t)out_blocks * 136) + (ab ? K.word : 0);
#pragma unroll 1
    for (int q = 0; q < out_blocks; ++q) {
        if (ab && K.main) o[(size_t)q * 17] = make_uint2(K.lo, K.hi);
        if (q + 1 < out_blocks) K.permute();
    }
}

int main(int argc, char **argv) {
    if (argc < 6) { printf("usage: keccak_wave in.bin out.bin N stride out_blocks [reps]\n"); return 2; }
    const size_t N = strtoull(argv[3], nullptr, 10), stride = strtoull(argv[4], nullptr, 10);
    const int out_blocks = atoi(argv[5]), reps = argc > 6 ? atoi(argv[6]) : 10;
    std::vector<uint8_t> h_in(N * stride + N * 4);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(h_in.data(), 1, h_in.size(), f) != h_in.size()) { printf("cannot read %s\n", argv[1]); return 2; }
    fclose(f);
    uint8_t *d_in, *d_out;
    const size_t out_bytes = N * (size_t)out_blocks * 136;
    CHECK(hipMalloc(&d_in, h_in.size()));
    CHECK(hipMalloc(&d_out, out_bytes));
    CHECK(hipMemcpy(d_in, h_in.data(), h_in.size(), hipMemcpyHostToDevice));
    CHECK(hipMemset(d_out, 0, out_bytes));
    const int *d_nb = reinterpret_cast<const int *>(d_in + N * stride);
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0));
    CHECK(hipEventCreate(&t1));
    for (int W : {1, 2, 4}) {
        for (int pass = 0; pass < 2; ++pass) {
            if (pass) CHECK(hipEventRecord(t0));
            for (int r = 0; r < (pass ? reps : 2); ++r) {
                if (W == 1) hipLaunchKernelGGL(shake_wave<1>, dim3((unsigned)N), dim3(64), 0, 0, d_in, stride, d_nb, N, out_blocks, d_out);
                else if (W == 2) hipLaunchKernelGGL(shake_wave<2>, dim3((unsigned)((N + 1) / 2)), dim3(128), 0, 0, d_in, stride, d_nb, N, out_blocks, d_out);
                else hipLaunchKernelGGL(shake_wave<4>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, 0, d_in, stride, d_nb, N, out_blocks, d_out);
            }
            CHECK(hipGetLastError());
            if (pass) { CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1)); } else CHECK(hipDeviceSynchronize());
        }
        float ms;
        CHECK(hipEventElapsedTime(&ms, t0, t1));
        printf("N=%zu waves/workgroup=%d: %.1f us per launch\n", N, W, ms * 1e3 / reps);
    }
    std::vector<uint8_t> h_out(out_bytes);
    CHECK(hipMemcpy(h_out.data(), d_out, out_bytes, hipMemcpyDeviceToHost));
    f = fopen(argv[2], "wb");
    if (!f || fwrite(h_out.data(), 1, out_bytes, f) != out_bytes) { printf("cannot write %s\n", argv[2]); return 2; }
    fclose(f);
    return 0;
}
