// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the ACCESS PATTERN of the step kernels (round 6; MI355X_MICROARCH.md: "calibrate on a known byte count in your own
// access pattern before trusting an absolute"): one wavefront per workgroup owns one unit = one row of W floats in each of three arrays (lane k reads element k: 4 B per lane,
// rows of 172 B for W = 43 -- not a multiple of a cache line), copies them to three output arrays and writes a 227-float record.  Variants: unit = blockIdx (neighbouring rows
// on different XCDs, each with its own L2) against the XCD-aware unit of dm_device.h dm_wg_unit(), and W = 43 against W = 48 (rows of three whole 64-B lines).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o f -- /tmp/fetch_calib      (and once more with WRITE_SIZE)
// The kernel NAME carries the variant; bytes per launch: read 3 * G * W * 4, written 3 * G * W * 4 + G * 227 * 4.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W, bool XCD>
__global__ void __launch_bounds__(64) k_calib(const float* a, const float* b, const float* c, float* oa, float* ob, float* oc, float* rec) {
    const int bi = blockIdx.x, G = gridDim.x, l = threadIdx.x;
    const int u = XCD ? (bi & 7) * (G >> 3) + (bi >> 3) : bi;
    float x = 0, y = 0, z = 0;
    if (l < W) { x = a[(size_t)u * W + l]; y = b[(size_t)u * W + l]; z = c[(size_t)u * W + l]; }
    float s = x + y + z;
    for (int i = 0; i < 2000; ++i) s = s * 1.0000001f + 1e-9f;            // a step kernel holds its rows for a while: the line is long gone from L1 when it is written
    if (l < W) { oa[(size_t)u * W + l] = x + 1; ob[(size_t)u * W + l] = y + 1; oc[(size_t)u * W + l] = z + s * 1e-30f; }
    for (int k = l; k < 227; k += 64) rec[(size_t)u * 227 + k] = s;
}
int main() {
    const int G = 2048;
    float *a, *b, *c, *oa, *ob, *oc, *rec;
    const size_t n = (size_t)G * 48;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4); hipMalloc(&oa, n * 4); hipMalloc(&ob, n * 4); hipMalloc(&oc, n * 4); hipMalloc(&rec, (size_t)G * 227 * 4);
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4); hipMemset(c, 0, n * 4);
    for (int rep = 0; rep < 6; ++rep) {
        // the outputs of one launch are the inputs of the next, like the env record of consecutive control steps
        hipLaunchKernelGGL((k_calib<43, false>), dim3(G), dim3(64), 0, 0, a, b, c, oa, ob, oc, rec);
        hipLaunchKernelGGL((k_calib<43, false>), dim3(G), dim3(64), 0, 0, oa, ob, oc, a, b, c, rec);
    }
    for (int rep = 0; rep < 6; ++rep) {
        hipLaunchKernelGGL((k_calib<43, true>), dim3(G), dim3(64), 0, 0, a, b, c, oa, ob, oc, rec);
        hipLaunchKernelGGL((k_calib<43, true>), dim3(G), dim3(64), 0, 0, oa, ob, oc, a, b, c, rec);
    }
    for (int rep = 0; rep < 6; ++rep) {
        hipLaunchKernelGGL((k_calib<48, false>), dim3(G), dim3(64), 0, 0, a, b, c, oa, ob, oc, rec);
        hipLaunchKernelGGL((k_calib<48, false>), dim3(G), dim3(64), 0, 0, oa, ob, oc, a, b, c, rec);
    }
    for (int rep = 0; rep < 6; ++rep) {
        hipLaunchKernelGGL((k_calib<48, true>), dim3(G), dim3(64), 0, 0, a, b, c, oa, ob, oc, rec);
        hipLaunchKernelGGL((k_calib<48, true>), dim3(G), dim3(64), 0, 0, oa, ob, oc, a, b, c, rec);
    }
    hipDeviceSynchronize();
    printf("G %d: W 43 reads %zu B, writes %zu B per launch; W 48 reads %zu B, writes %zu B\n", G, (size_t)3 * G * 43 * 4, (size_t)3 * G * 43 * 4 + (size_t)G * 227 * 4,
           (size_t)3 * G * 48 * 4, (size_t)3 * G * 48 * 4 + (size_t)G * 227 * 4);
    return 0;
}
