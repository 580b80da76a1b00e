// Register layout of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products):
// D[v][lane] = A[la] * B[lb] -- which (la, lb)?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const float a = 1.0f + l, b = 128.0f + l;
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[v * 64 + l] = c[v];
}
int main() {
    float* d; hipMalloc(&d, 1024);
    probe<<<1, 64>>>(d);
    float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int v = 0; v < 4; ++v)
        for (int l = 0; l < 64; ++l) {
            int found = 0;
            for (int la = 0; la < 64 && !found; ++la)
                for (int lb = 0; lb < 64; ++lb)
                    if ((1.0f + la) * (128.0f + lb) == h[v * 64 + l]) {
                        if (l < 12 || l > 58) printf("D[vgpr %d][lane %2d] = A[lane %2d] * B[lane %2d]\n", v, l, la, lb);
                        found = 1; break;
                    }
            if (!found) printf("D[%d][%d] = %g ?\n", v, l, h[v * 64 + l]);
        }
    return 0;
}
