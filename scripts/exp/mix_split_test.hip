// v_fma_mixlo/hi_f16 split against the plain C++ arithmetic, three asm forms (scripts/exp: experiments, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef _Float16 v2h_t __attribute__((ext_vector_type(2)));
__device__ void ref_split(float x0, float x1, unsigned& h, unsigned& l) {
    const v2h_t hh = {(_Float16)x0, (_Float16)x1};
    const v2h_t ll = {(_Float16)(x0 - (float)hh[0]), (_Float16)(x1 - (float)hh[1])};
    h = __builtin_bit_cast(unsigned, hh); l = __builtin_bit_cast(unsigned, ll);
}
__global__ void k(const float* x, unsigned* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x0 = x[2 * i], x1 = x[2 * i + 1];
    unsigned h, l;
    ref_split(x0, x1, h, l);
    unsigned l1;
    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                 : "=&v"(l1) : "v"(h), "v"(x0), "v"(x1));
    unsigned l2 = __float_as_uint(x0);
    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                 : "+v"(l2) : "v"(h), "v"(x1));
    float m1 = -1.0f;
    asm volatile("" : "+s"(m1));
    unsigned l3;
    asm volatile("v_fma_mixlo_f16 %0, %1, %4, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %4, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                 : "=&v"(l3) : "v"(h), "v"(x0), "v"(x1), "s"(m1));
    out[5 * i] = h; out[5 * i + 1] = l; out[5 * i + 2] = l1; out[5 * i + 3] = l2; out[5 * i + 4] = l3;
}
int main() {
    const int n = 1 << 16;
    float* hx = (float*)malloc(2 * n * sizeof(float));
    srand(1);
    for (int i = 0; i < 2 * n; ++i) {
        float m = (float)rand() / RAND_MAX * 2.f - 1.f;
        int e = rand() % 40 - 30;
        hx[i] = ldexpf(m, e);
        if (i % 97 == 0) hx[i] = 0.f;
    }
    float* dx; unsigned* dout;
    hipMalloc(&dx, 2 * n * sizeof(float)); hipMalloc(&dout, 5 * n * sizeof(unsigned));
    hipMemcpy(dx, hx, 2 * n * sizeof(float), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    unsigned* ho = (unsigned*)malloc(5 * n * sizeof(unsigned));
    hipMemcpy(ho, dout, 5 * n * sizeof(unsigned), hipMemcpyDeviceToHost);
    int bad[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int v = 0; v < 3; ++v)
            if (ho[5 * i + 2 + v] != ho[5 * i + 1]) {
                if (bad[v]++ < 5) printf("form %d: x = (%g, %g) hi %08x ref lo %08x got %08x\n", v + 1, hx[2 * i], hx[2 * i + 1], ho[5 * i], ho[5 * i + 1], ho[5 * i + 2 + v]);
            }
    printf("mismatches of %d pairs: form1 (=&v) %d, form2 (tied) %d, form3 (sgpr -1) %d\n", n, bad[0], bad[1], bad[2]);
    return 0;
}
