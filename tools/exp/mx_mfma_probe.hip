// mx_mfma_probe.hip — which lane holds which (row, k) of the operands of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 and fp4 e2m1),
// and which k block a lane's E8M0 scale byte applies to?  Standalone probe (no torch):
//     hipcc --offload-arch=gfx950 -O2 tools/exp/mx_mfma_probe.hip -o tools/exp/mx_mfma_probe && tools/exp/mx_mfma_probe
// Random small-integer A [32 x 64] and B [64 x 32] are packed under a list of candidate layouts; the candidate whose device result
// equals the host product names the layout.  C/D is the 32x32 map of every gfx950 MFMA: col = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int FMT>      // 0 = fp8 e4m3, 4 = fp4 e2m1
__global__ void probe(const v8i* a, const v8i* b, const int* sa, const int* sb, float* out) {
    v16f c = {0};
    const int l = threadIdx.x;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, FMT, FMT, 0, sa[l], 0, sb[l]);
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = c[i];
}

static unsigned char e4m3(int v) {          // small integers only
    static const unsigned char tab[5] = {0x00, 0x38, 0x40, 0x44, 0x48};
    return (unsigned char)((v < 0 ? 0x80 : 0) | tab[abs(v)]);
}
static unsigned char e2m1(int v) {          // 0, 1, 2, 3, 4 -> codes 0, 2, 4, 5, 6  (0.5 = 1, 1.5 = 3, 6 = 7)
    static const unsigned char tab[5] = {0, 2, 4, 5, 6};
    return (unsigned char)((v < 0 ? 8 : 0) | tab[abs(v)]);
}

// candidate k index of element j (0..31 for fp8; 0..31 for fp4 too: 32 nibbles in the first 16 bytes) held by lane group g = lane >> 5
static int k_of(int cand, int g, int j) {
    switch (cand) {
        case 0: return g * 32 + j;                                   // contiguous halves
        case 1: return (j >> 4) * 32 + g * 16 + (j & 15);            // two groups of 16
        case 2: return (j >> 3) * 16 + g * 8 + (j & 7);              // four groups of 8
        case 3: return j * 2 + g;                                    // interleaved
        default: return (j >> 2) * 8 + g * 4 + (j & 3);              // eight groups of 4
    }
}

int main() {
    const int NC = 5;
    std::vector<int> A(32 * 64), B(64 * 32);
    srand(7);
    for (auto& x : A) x = rand() % 9 - 4;
    for (auto& x : B) x = rand() % 9 - 4;
    v8i *da, *db; int *dsa, *dsb; float* dout;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dout, 64 * 16 * 4);
    std::vector<float> out(64 * 16);
    for (int fmt = 0; fmt <= 4; fmt += 4) {
        for (int cand = 0; cand < NC; ++cand) {
            for (int scale_test = 0; scale_test < 3; ++scale_test) {
                std::vector<unsigned char> pa(64 * 32, 0), pb(64 * 32, 0);
                std::vector<int> sa(64), sb(64);
                for (int l = 0; l < 64; ++l) {
                    const int g = l >> 5, rc = l & 31;
                    for (int j = 0; j < 32; ++j) {
                        const int k = k_of(cand, g, j);
                        if (fmt == 0) { pa[l * 32 + j] = e4m3(A[rc * 64 + k]); pb[l * 32 + j] = e4m3(B[k * 32 + rc]); }
                        else {
                            pa[l * 32 + j / 2] |= (unsigned char)(e2m1(A[rc * 64 + k]) << ((j & 1) * 4));
                            pb[l * 32 + j / 2] |= (unsigned char)(e2m1(B[k * 32 + rc]) << ((j & 1) * 4));
                        }
                    }
                    // scale tests: 0 = all 2^0; 1 = A scale 2^1 in the lanes of group 1 only; 2 = B scale 2^2 in the lanes of group 0 only
                    sa[l] = (scale_test == 1 && g == 1) ? 128 : 127;
                    sb[l] = (scale_test == 2 && g == 0) ? 129 : 127;
                }
                hipMemcpy(da, pa.data(), 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, pb.data(), 64 * 32, hipMemcpyHostToDevice);
                hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
                if (fmt == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
                else hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
                hipMemcpy(out.data(), dout, 64 * 16 * 4, hipMemcpyDeviceToHost);
                // host reference under "the scale of lane group g applies to the k values that group holds"
                int bad = 0; double worst = 0;
                for (int l = 0; l < 64; ++l)
                    for (int r = 0; r < 16; ++r) {
                        const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                        double ref = 0;
                        for (int g = 0; g < 2; ++g)
                            for (int j = 0; j < 32; ++j) {
                                const int k = k_of(cand, g, j);
                                double w = 1.0;
                                if (scale_test == 1 && g == 1) w = 2.0;
                                if (scale_test == 2 && g == 0) w = 4.0;
                                ref += w * A[row * 64 + k] * B[k * 32 + col];
                            }
                        const double d = fabs(ref - out[l * 16 + r]);
                        if (d > 1e-3) ++bad;
                        if (d > worst) worst = d;
                    }
                printf("fmt=%s layout candidate %d scale test %d: %s (%d / 1024 off, worst %.1f)\n", fmt ? "fp4" : "fp8", cand, scale_test,
                       bad ? "no" : "MATCH", bad, worst);
            }
        }
    }
    // Which elements does a lane group's scale byte apply to?  One-hot A element (row 0, lane group g, byte j) against all-ones B with
    // the A scale 2^1 in lane group 1 only (and vice versa for B): the result is that element's effective scale.
    for (int which = 0; which < 2; ++which) {
        printf("%s scale = 2 in the lanes of group 1: effective scale of element (g, j), fp8\n", which ? "B" : "A");
        for (int g = 0; g < 2; ++g) {
            printf("  g=%d:", g);
            for (int j = 0; j < 32; ++j) {
                std::vector<unsigned char> pa(64 * 32, 0), pb(64 * 32, 0);
                std::vector<int> sa(64, 127), sb(64, 127);
                for (int l = 0; l < 64; ++l)
                    for (int jj = 0; jj < 32; ++jj) (which ? pa : pb)[l * 32 + jj] = 0x38;          // the other operand: all ones
                (which ? pb : pa)[(g * 32 + 0) * 32 + j] = 0x38;                                   // one element of row / column 0
                for (int l = 32; l < 64; ++l) (which ? sb : sa)[l] = 128;
                hipMemcpy(da, pa.data(), 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, pb.data(), 64 * 32, hipMemcpyHostToDevice);
                hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
                hipMemcpy(out.data(), dout, 64 * 16 * 4, hipMemcpyDeviceToHost);
                // C[0][0] is lane 0 reg 0 when A is one-hot in row 0; when B is one-hot in column 0 it is lane 0 reg 0 as well
                printf(" %.0f", out[0]);
            }
            printf("\n");
        }
    }
    // the other bytes of the scale register (opsel = 0 reads byte 0): scale byte 1 set instead of byte 0 must change nothing
    return 0;
}
