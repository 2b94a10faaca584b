// Lane -> element maps of the MFMA shapes round 4 uses, found by one-hot probing (no ISA document in this image):
//   v_mfma_f64_4x4x4_4b_f64   (four independent 4x4x4 products per instruction: the Schur pair assembly, csrc/lm.hip)
//   v_mfma_f64_16x16x4_f64    (the tile solver's trailing update; listed for reference)
//   v_mfma_i32_32x32x32_i8    (Hamming distance as (256 - a.b) / 2 over +-1 bytes, csrc/hamming.hip)
//   v_mfma_i32_16x16x64_i8
// For every operand slot (lane, element) of A the probe sets that slot to 1 with B all ones: the D slots that light up are the slot's ROW;
// likewise for B (its COLUMN); A one-hot x B one-hot lights a D slot iff both carry the same k.  The tables are printed as formulas' raw material.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// MODE 0: f64 4x4x4 (D: 1 double per lane), 1: f64 16x16x4 (D: 4 per lane), 2: i8 32x32x32 (A/B 16 bytes per lane, D 16 ints), 3: i8 16x16x64 (A/B 16 bytes, D 4 ints)
template <int MODE>
__global__ void k(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B, double* __restrict__ D) {
    const int l = threadIdx.x;
    if (MODE == 0) {
        double a = A[l], b = B[l];
        double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        D[l] = d;
    } else if (MODE == 1) {
        double a = A[l], b = B[l];
        v4d c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; r++) D[l * 4 + r] = c[r];
    } else if (MODE == 2) {
        v4i a, b;
        memcpy(&a, A + 16 * l, 16); memcpy(&b, B + 16 * l, 16);
        v16i c;
        for (int r = 0; r < 16; r++) c[r] = 0;
        c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
        for (int r = 0; r < 16; r++) D[l * 16 + r] = c[r];
    } else {
        v4i a, b;
        memcpy(&a, A + 16 * l, 16); memcpy(&b, B + 16 * l, 16);
        v4i c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; r++) D[l * 4 + r] = c[r];
    }
}

template <int MODE> void probe(const char* name, int epl /* A/B elements per lane */, int dpl /* D elements per lane */) {
    const int nA = 64 * epl, nD = 64 * dpl;
    unsigned char *dA, *dB; double* dD;
    hipMalloc(&dA, nA); hipMalloc(&dB, nA); hipMalloc(&dD, nD * 8);
    std::vector<unsigned char> hA(nA), hB(nA);
    std::vector<double> hD(nD);
    auto run = [&]() {
        hipMemcpy(dA, hA.data(), nA, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), nA, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(hD.data(), dD, nD * 8, hipMemcpyDeviceToHost);
    };
    printf("==== %s: %d A/B elements per lane, %d D elements per lane\n", name, epl, dpl);
    // rows: A one-hot, B ones
    std::vector<std::vector<int>> rowSet(nA), colSet(nA);
    for (int s = 0; s < nA; s++) {
        std::fill(hA.begin(), hA.end(), 0); std::fill(hB.begin(), hB.end(), 1); hA[s] = 1; run();
        for (int d = 0; d < nD; d++) if (hD[d] != 0) rowSet[s].push_back(d);
        std::fill(hA.begin(), hA.end(), 1); std::fill(hB.begin(), hB.end(), 0); hB[s] = 1; run();
        for (int d = 0; d < nD; d++) if (hD[d] != 0) colSet[s].push_back(d);
    }
    // label rows / columns by their D-slot sets
    std::vector<std::vector<int>> rows, cols;
    std::vector<int> rowOf(nA), colOf(nA);
    for (int s = 0; s < nA; s++) {
        int r = -1; for (size_t q = 0; q < rows.size(); q++) if (rows[q] == rowSet[s]) r = (int)q;
        if (r < 0) { rows.push_back(rowSet[s]); r = (int)rows.size() - 1; }
        rowOf[s] = r;
        int c = -1; for (size_t q = 0; q < cols.size(); q++) if (cols[q] == colSet[s]) c = (int)q;
        if (c < 0) { cols.push_back(colSet[s]); c = (int)cols.size() - 1; }
        colOf[s] = c;
    }
    printf("distinct rows %zu (each lights %zu D slots), distinct columns %zu (each lights %zu)\n", rows.size(), rows[0].size(), cols.size(), cols[0].size());
    // k classes: A slot s pairs with B slot 0's class ...  (A one-hot s, B one-hot t => nonzero iff same k and compatible block)
    // cheap version: for A slot s find the B slots of lane-element order that give a nonzero
    std::vector<int> kOfA(nA, -1), kOfB(nA, -1);
    int nk = 0;
    for (int t = 0; t < nA; t++) {
        if (kOfB[t] >= 0) continue;
        // new k class seeded by B slot t: every A slot that multiplies with it, then every B slot that multiplies with one of those
        std::fill(hB.begin(), hB.end(), 0); hB[t] = 1; std::fill(hA.begin(), hA.end(), 1); run();   // (only to skip empty)
        const int kc = nk++;
        kOfB[t] = kc;
        int aSeed = -1;
        for (int s = 0; s < nA; s++) {
            std::fill(hA.begin(), hA.end(), 0); hA[s] = 1; std::fill(hB.begin(), hB.end(), 0); hB[t] = 1; run();
            bool nz = false; for (int d = 0; d < nD; d++) nz |= hD[d] != 0;
            if (nz) { kOfA[s] = kc; if (aSeed < 0) aSeed = s; }
        }
        if (aSeed >= 0)
            for (int u = 0; u < nA; u++) {
                if (kOfB[u] >= 0) continue;
                std::fill(hA.begin(), hA.end(), 0); hA[aSeed] = 1; std::fill(hB.begin(), hB.end(), 0); hB[u] = 1; run();
                bool nz = false; for (int d = 0; d < nD; d++) nz |= hD[d] != 0;
                if (nz) kOfB[u] = kc;
            }
    }
    printf("k classes %d\n", nk);
    printf("A slot (lane,elem) -> row label, k class:\n");
    for (int l = 0; l < 64; l++) { printf("  lane %2d:", l); for (int e = 0; e < epl; e++) printf(" (r%d,k%d)", rowOf[l * epl + e], kOfA[l * epl + e]); printf("\n"); }
    printf("B slot (lane,elem) -> column label, k class:\n");
    for (int l = 0; l < 64; l++) { printf("  lane %2d:", l); for (int e = 0; e < epl; e++) printf(" (c%d,k%d)", colOf[l * epl + e], kOfB[l * epl + e]); printf("\n"); }
    printf("D slot (lane,reg) -> (row label, column label):\n");
    std::vector<int> dRow(nD, -1), dCol(nD, -1);
    for (size_t r = 0; r < rows.size(); r++) for (int d : rows[r]) dRow[d] = (int)r;
    for (size_t c = 0; c < cols.size(); c++) for (int d : cols[c]) dCol[d] = (int)c;
    for (int l = 0; l < 64; l++) { printf("  lane %2d:", l); for (int r = 0; r < dpl; r++) printf(" (r%d,c%d)", dRow[l * dpl + r], dCol[l * dpl + r]); printf("\n"); }
    hipFree(dA); hipFree(dB); hipFree(dD);
}
int main() {
    probe<0>("v_mfma_f64_4x4x4_4b_f64", 1, 1);
    probe<1>("v_mfma_f64_16x16x4_f64", 1, 4);
    probe<2>("v_mfma_i32_32x32x32_i8", 16, 16);
    probe<3>("v_mfma_i32_16x16x64_i8", 16, 4);
    return 0;
}
