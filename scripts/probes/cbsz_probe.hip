// cbsz_probe.hip -- what CBSZ / ABID do on v_mfma_f32_4x4x1_16b_f32 (gfx950): with CBSZ = 4 the A operand of block ABID is broadcast
// to all 16 blocks, so ONE VGPR can hold the A operands of 16 different MFMAs (lanes 4a..4a+3 = the 4 neurons of MFMA a).
// build: hipcc --offload-arch=gfx950 -O2 scripts/probes/cbsz_probe.hip -o /tmp/cbsz_probe && /tmp/cbsz_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    // A = lane id, B = 1 + lane/1000: D[i] at lane l = A(block, i) * B(l)
    f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)lane, 1.0f, c, CBSZ, ABID, 0);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = d[i];
}

template <int CBSZ, int ABID>
int run(float* d_out) {
    float h[256];
    hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, d_out);
    (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            // expectation: groups of 2^CBSZ blocks share the A of block (group base + ABID)
            const int blk = l >> 2, grp = blk >> CBSZ << CBSZ;
            const float want = (float)(4 * (grp + ABID) + i);
            if (h[l * 4 + i] != want) ++bad;
        }
    printf("cbsz=%d abid=%2d: lane0 D = %g %g %g %g | lane 37 D = %g %g %g %g | mismatches vs model: %d\n", CBSZ, ABID, h[0], h[1], h[2], h[3],
           h[37 * 4], h[37 * 4 + 1], h[37 * 4 + 2], h[37 * 4 + 3], bad);
    return bad;
}

int main() {
    float* d;
    (void)hipMalloc(&d, 256 * sizeof(float));
    int bad = 0;
    bad += run<0, 0>(d);
    bad += run<4, 0>(d);
    bad += run<4, 5>(d);
    bad += run<4, 15>(d);
    bad += run<2, 3>(d);
    bad += run<1, 1>(d);
    printf(bad ? "MODEL WRONG\n" : "model confirmed: CBSZ=4 broadcasts block ABID's A to all 16 blocks\n");
    return bad != 0;
}
