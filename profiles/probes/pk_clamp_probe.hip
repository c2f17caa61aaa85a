// Does the clamp output modifier act on the packed float32 VALU operations of gfx950?  (hipcc assembles it on all three.)
//   hipcc --offload-arch=gfx950 -O2 profiles/probes/pk_clamp_probe.hip -o /tmp/pk_clamp && /tmp/pk_clamp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const f32x2 *x, f32x2 *y) {
    const f32x2 v = x[threadIdx.x], one = {1.0f, 1.0f}, zero = {0.0f, 0.0f};
    f32x2 a, b, c;
    asm volatile("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(a) : "v"(v), "v"(one));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(b) : "v"(v), "v"(one), "v"(zero));
    asm volatile("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(c) : "v"(v), "v"(zero));
    float d;
    asm volatile("v_mul_f32_e64 %0, %1, 1.0 clamp" : "=v"(d) : "v"(v[0]));
    y[4 * threadIdx.x] = a;
    y[4 * threadIdx.x + 1] = b;
    y[4 * threadIdx.x + 2] = c;
    y[4 * threadIdx.x + 3] = f32x2{d, 0.0f};
}
int main() {
    f32x2 h[4] = {{-2.5f, 0.25f}, {0.75f, 3.0f}, {-0.0f, 1.0f}, {1e-3f, -1e-3f}}, *dx, *dy, out[16];
    hipMalloc(&dx, sizeof(h));
    hipMalloc(&dy, sizeof(out));
    hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 4>>>(dx, dy);
    hipMemcpy(out, dy, sizeof(out), hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; i++)
        printf("in (%g, %g): pk_mul clamp (%g, %g)  pk_fma clamp (%g, %g)  pk_add clamp (%g, %g)  v_mul clamp %g\n", h[i][0], h[i][1], out[4 * i][0],
               out[4 * i][1], out[4 * i + 1][0], out[4 * i + 1][1], out[4 * i + 2][0], out[4 * i + 2][1], out[4 * i + 3][0]);
    return 0;
}
