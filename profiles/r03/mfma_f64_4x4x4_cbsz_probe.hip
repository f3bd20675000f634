// Does v_mfma_f64_4x4x4_4b_f64 honour the A-broadcast controls (CBSZ / ABID) on gfx950?  With CBSZ = 2 block ABID's
// A values should feed all four blocks: D_b = A_abid . B_b.  Lane maps (profiles/r02/mfma_f64_4x4x4_probe.hip):
// A_b[i][k] @ lane 16 k + 4 b + i, B_b[k][j] @ lane 16 k + 4 b + j, D_b[i][j] @ lane 16 i + 4 b + j.
//   hipcc --offload-arch=gfx950 -O3 profiles/r03/mfma_f64_4x4x4_cbsz_probe.hip -o cbsz_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int ABID>
__global__ void k(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d[ABID * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 2, ABID, 0);
}
int main() {
    double ha[64], hb[64], hd[256], *da, *db, *dd;
    for (int l = 0; l < 64; ++l) {
        ha[l] = 1.0 + 0.37 * l + 0.01 * l * l;
        hb[l] = 2.0 - 0.11 * l + 0.003 * l * l;
    }
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 2048);
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    k<0><<<1, 64>>>(da, db, dd); k<1><<<1, 64>>>(da, db, dd); k<2><<<1, 64>>>(da, db, dd); k<3><<<1, 64>>>(da, db, dd);
    hipMemcpy(hd, dd, 2048, hipMemcpyDeviceToHost);
    for (int abid = 0; abid < 4; ++abid) {
        // which source block (if any) explains every output block?
        for (int blk = 0; blk < 4; ++blk) {
            printf("CBSZ=2 ABID=%d block %d:", abid, blk);
            for (int src = 0; src < 4; ++src) {
                double worst = 0;
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) {
                        double want = 0;
                        for (int kk = 0; kk < 4; ++kk) want += ha[16 * kk + 4 * src + i] * hb[16 * kk + 4 * blk + j];
                        worst = fmax(worst, fabs(want - hd[abid * 64 + 16 * i + 4 * blk + j]));
                    }
                printf("  A_%d.B_%d err %.1e", src, blk, worst);
            }
            printf("\n");
        }
    }
    return 0;
}
