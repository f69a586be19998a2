// mfma8x8.hip — VERDICT r1 item 7 / north_star "MFMA only where it pays": the 8x8 x 8x8 fp64 products of the Riccati stage
// (P A, A^T (P A)) as a dependent chain on ONE wavefront, three ways:
//   fma     what k_riccati1 does: lane (i, g) owns element (i, g), holds row i of P in registers, reads column g of A from
//           LDS, 8 dependent v_fma_f64, writes its element to LDS and gathers the row again (8 ds_read)
//   mfma16  v_mfma_f64_16x16x4_f64: two chains per wavefront stacked as [P1; P2] (16x8) x [A1 | A2] (8x16), the diagonal
//           8x8 blocks of the 16x16 result are the two products; result -> LDS -> next A operand (2 reads per lane)
//   mfma4   v_mfma_f64_4x4x4_4b_f64: four 4x4x4 blocks per instruction, an 8x8x8 product = 2 instructions (K halves) over
//           the 2x2 output blocks; operand layout found by probing (printed), result -> LDS -> next A operand
// Each variant runs NREP chained products P <- P A (A orthogonal, so P stays bounded) and is checked against the CPU.
// Prints shader cycles (s_memtime) per 8x8x8 product.  Build: hipcc -O3 --offload-arch=gfx950 mfma8x8.hip -o mfma8x8
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
#define WSYNC()                                            \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

// ---- fma: one chain per wavefront
__global__ void __launch_bounds__(64) k_fma(const double* __restrict__ P0, const double* __restrict__ A, double* __restrict__ out, long long* cyc, int nrep) {
  __shared__ double lA[64], lP[64];
  const int lane = threadIdx.x, i = lane >> 3, g = lane & 7;
  lA[lane] = A[lane];
  double Prow[8];
  for (int l = 0; l < 8; l++) Prow[l] = P0[i * 8 + l];
  WSYNC();
  const long long t0 = clock64();
  for (int r = 0; r < nrep; r++) {
    double Ag[8];
#pragma unroll
    for (int l = 0; l < 8; l++) Ag[l] = lA[l * 8 + g];
    double pa = 0.0;
#pragma unroll
    for (int l = 0; l < 8; l++) pa += Prow[l] * Ag[l];
    lP[i * 8 + g] = pa;
    WSYNC();
#pragma unroll
    for (int l = 0; l < 8; l++) Prow[l] = lP[i * 8 + l];
    WSYNC();
  }
  const long long t1 = clock64();
  out[lane] = Prow[g];
  if (lane == 0) cyc[0] = t1 - t0;
}

// ---- mfma16: two chains per wavefront.  A operand: lane l holds X[row l & 15][k = 4 c + (l >> 4)] for K chunk c;
// B operand: Y[k = 4 c + (l >> 4)][col l & 15]; D: col = l & 15, row = (l >> 4) + 4 reg.
__global__ void __launch_bounds__(64) k_mfma16(const double* __restrict__ P0, const double* __restrict__ A, double* __restrict__ out, long long* cyc, int nrep) {
  __shared__ double lP[16 * 8];  // [row of the stack][k]
  const int lane = threadIdx.x, r16 = lane & 15, q = lane >> 4;
  // stack: rows 0..7 = chain 1, rows 8..15 = chain 2 (same P0 and A for both: same numbers, twice the work)
  double a0 = P0[(r16 & 7) * 8 + q], a1 = P0[(r16 & 7) * 8 + 4 + q];
  const double b0 = A[q * 8 + (r16 & 7)], b1 = A[(4 + q) * 8 + (r16 & 7)];  // [A | A]: col r16 -> column r16 & 7
  WSYNC();
  const long long t0 = clock64();
  for (int r = 0; r < nrep; r++) {
    v4d d = {0.0, 0.0, 0.0, 0.0};
    d = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, d, 0, 0, 0);
    // diagonal blocks: rows 0..7 x cols 0..7 (chain 1), rows 8..15 x cols 8..15 (chain 2): row = q + 4 reg
    const int hi = r16 >> 3;  // this lane's column belongs to chain hi
    lP[(q + 8 * hi) * 8 + (r16 & 7)] = hi ? d[2] : d[0];
    lP[(q + 4 + 8 * hi) * 8 + (r16 & 7)] = hi ? d[3] : d[1];
    WSYNC();
    a0 = lP[r16 * 8 + q], a1 = lP[r16 * 8 + 4 + q];
    WSYNC();
  }
  const long long t1 = clock64();
  // element (i, g) of chain 1 to out[i * 8 + g]: lane (row i = r16 < 8, k = q) holds P[i][q], P[i][4 + q]
  if (r16 < 8) out[r16 * 8 + q] = a0, out[r16 * 8 + 4 + q] = a1;
  if (lane == 0) cyc[0] = t1 - t0;
}

// ---- probe of v_mfma_f64_4x4x4_4b: which (block, i, k) / (block, k, j) / (block, i, j) does a lane hold?
__global__ void k_probe4(double* res) {  // res[which][src lane][dst lane]
  const int lane = threadIdx.x;
  for (int s = 0; s < 64; s++) {
    double a = lane == s ? 1.0 : 0.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, 1.0, 0.0, 0, 0, 0);   // D = sum_k A[i][k] over lanes holding 1 in B
    res[(0 * 64 + s) * 64 + lane] = d;
    double b = lane == s ? 1.0 : 0.0;
    d = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, b, 0.0, 0, 0, 0);
    res[(1 * 64 + s) * 64 + lane] = d;
  }
}

// ---- mfma4: one chain per wavefront, layout parameters found by the probe and passed in:
// lane l: block bl = l / 16; A operand element (i = (l % 16) / ai_div % 4 ...) -> generic tables from the host
__global__ void __launch_bounds__(64) k_mfma4(const double* __restrict__ P0, const double* __restrict__ A, double* __restrict__ out, long long* cyc, int nrep,
                                              const int* __restrict__ tab) {
  // tab[0..63]: for this lane as A operand: block-row I*? encoded as (rowA * 8 + kA) for K half 0; +64: K half 1;
  // tab[128..191]: as B operand (kB * 8 + colB) for K half 0; +192 half 1; tab[256..319]: output element (row * 8 + col)
  __shared__ double lP[64];
  const int lane = threadIdx.x;
  const int ia0 = tab[lane], ia1 = tab[64 + lane], ib0 = tab[128 + lane], ib1 = tab[192 + lane], io = tab[256 + lane];
  double a0 = P0[ia0], a1 = P0[ia1];
  const double b0 = A[ib0], b1 = A[ib1];
  WSYNC();
  const long long t0 = clock64();
  for (int r = 0; r < nrep; r++) {
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, 0.0, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, d, 0, 0, 0);
    lP[io] = d;
    WSYNC();
    a0 = lP[ia0], a1 = lP[ia1];
    WSYNC();
  }
  const long long t1 = clock64();
  out[io] = lP[io];
  if (lane == 0) cyc[0] = t1 - t0;
}

int main() {
  const int nrep = 4000;
  std::vector<double> P(64), A(64), ref(64);
  // A = product of plane rotations (orthogonal, asymmetric), P = asymmetric
  for (int i = 0; i < 64; i++) A[i] = (i / 8 == i % 8) ? 1.0 : 0.0;
  for (int p = 0; p < 7; p++) {
    double c = cos(0.3 + 0.1 * p), s = sin(0.3 + 0.1 * p);
    for (int r = 0; r < 8; r++) {
      double x = A[r * 8 + p], y = A[r * 8 + p + 1];
      A[r * 8 + p] = c * x - s * y, A[r * 8 + p + 1] = s * x + c * y;
    }
  }
  for (int i = 0; i < 64; i++) P[i] = 0.1 * (i % 8) - 0.07 * (i / 8) + (i == 13 ? 1.0 : 0.0);
  ref = P;
  for (int r = 0; r < nrep; r++) {
    std::vector<double> n(64, 0.0);
    for (int i = 0; i < 8; i++)
      for (int j = 0; j < 8; j++) {
        double s = 0.0;
        for (int l = 0; l < 8; l++) s += ref[i * 8 + l] * A[l * 8 + j];
        n[i * 8 + j] = s;
      }
    ref = n;
  }
  double *dP, *dA, *dout, *dres;
  long long* dc;
  int* dtab;
  hipMalloc(&dP, 512), hipMalloc(&dA, 512), hipMalloc(&dout, 512), hipMalloc(&dc, 8), hipMalloc(&dres, 2 * 64 * 64 * 8), hipMalloc(&dtab, 320 * 4);
  hipMemcpy(dP, P.data(), 512, hipMemcpyHostToDevice), hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice);
  auto report = [&](const char* name, int chains) {
    std::vector<double> o(64);
    long long c;
    hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost), hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 64; i++) err = fmax(err, fabs(o[i] - ref[i]));
    printf("%-7s %8.1f cycles per 8x8x8 product (%d chain%s per wavefront, %d chained products, %lld cycles); max |err| vs CPU %.2e\n", name,
           (double)c / nrep / chains, chains, chains > 1 ? "s" : "", nrep, c, err);
  };
  for (int rep = 0; rep < 2; rep++) {  // second pass: warm
    hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, 0, dP, dA, dout, dc, nrep);
    hipDeviceSynchronize();
    if (rep) report("fma", 1);
    hipLaunchKernelGGL(k_mfma16, dim3(1), dim3(64), 0, 0, dP, dA, dout, dc, nrep);
    hipDeviceSynchronize();
    if (rep) report("mfma16", 2);
  }
  // probe the 4x4x4_4b layout
  hipLaunchKernelGGL(k_probe4, dim3(1), dim3(64), 0, 0, dres);
  hipDeviceSynchronize();
  std::vector<double> res(2 * 64 * 64);
  hipMemcpy(res.data(), dres, res.size() * 8, hipMemcpyDeviceToHost);
  // A-lane s feeds output lanes {d}: same block, same row i; B-lane s feeds output lanes with same block, same column j.
  // From the sets: rowA[s] = the common row, found by intersecting with B's sets (an output lane is (row, col)).
  // Build: for each output lane d: the set of A lanes feeding it (4 lanes: k = 0..3) and of B lanes (4 lanes).
  std::vector<int> tab(320, 0);
  bool ok = true;
  // group output lanes by identical A-feeder sets -> same (block, row); by identical B-feeder sets -> same (block, col)
  auto feeders = [&](int which, int d) { unsigned long long m = 0; for (int s = 0; s < 64; s++) if (res[(which * 64 + s) * 64 + d] != 0.0) m |= 1ull << s; return m; };
  // blocks: lanes whose A feeders intersect
  int block[64], row[64], col[64];
  for (int d = 0; d < 64; d++) block[d] = row[d] = col[d] = -1;
  int nb = 0;
  for (int d = 0; d < 64; d++) {
    if (block[d] >= 0) continue;
    // all lanes sharing a row or a column transitively
    std::vector<int> q{d};
    block[d] = nb;
    for (size_t h = 0; h < q.size(); h++)
      for (int e = 0; e < 64; e++)
        if (block[e] < 0 && (feeders(0, e) == feeders(0, q[h]) || feeders(1, e) == feeders(1, q[h]))) block[e] = nb, q.push_back(e);
    nb++;
  }
  printf("4x4x4_4b probe: %d blocks;", nb);
  for (int b = 0; b < nb && b < 4; b++) {
    int nr = 0, nc = 0;
    std::vector<unsigned long long> rs, cs;
    for (int d = 0; d < 64; d++)
      if (block[d] == b) {
        unsigned long long fa = feeders(0, d), fb = feeders(1, d);
        size_t ir = 0; while (ir < rs.size() && rs[ir] != fa) ir++;
        if (ir == rs.size()) rs.push_back(fa);
        size_t ic = 0; while (ic < cs.size() && cs[ic] != fb) ic++;
        if (ic == cs.size()) cs.push_back(fb);
        row[d] = (int)ir, col[d] = (int)ic;
        nr = (int)rs.size(), nc = (int)cs.size();
      }
    if (nr != 4 || nc != 4) ok = false;
  }
  printf(" lane -> (block,row,col) of D:");
  for (int d = 0; d < 64; d += 5) printf(" %d:(%d,%d,%d)", d, block[d], row[d], col[d]);
  printf("\n");
  if (ok && nb == 4) {
    // A-operand lane s feeds the outputs of ONE (block, row): its k index = position among the 4 feeders of that output, ordered by lane
    int ablock[64], arow[64], ak[64], bblock[64], bcol[64], bk[64];
    for (int s = 0; s < 64; s++) {
      ablock[s] = arow[s] = ak[s] = bblock[s] = bcol[s] = bk[s] = -1;
      for (int d = 0; d < 64; d++) {
        if (res[(0 * 64 + s) * 64 + d] != 0.0 && ablock[s] < 0) {
          ablock[s] = block[d], arow[s] = row[d];
          unsigned long long f = feeders(0, d); int k = 0; for (int e = 0; e < s; e++) if (f >> e & 1) k++; ak[s] = k;
        }
        if (res[(1 * 64 + s) * 64 + d] != 0.0 && bblock[s] < 0) {
          bblock[s] = block[d], bcol[s] = col[d];
          unsigned long long f = feeders(1, d); int k = 0; for (int e = 0; e < s; e++) if (f >> e & 1) k++; bk[s] = k;
        }
      }
    }
    // the k index of A feeders and of B feeders of one output must pair up: check with a random product below.
    // 8x8x8 = 2x2 output blocks (I, J) = block b -> I = b >> 1, J = b & 1; K halves in two instructions
    for (int l = 0; l < 64; l++) {
      int I = ablock[l] >> 1, J = bblock[l] & 1;
      tab[l] = (4 * I + arow[l]) * 8 + ak[l], tab[64 + l] = (4 * I + arow[l]) * 8 + 4 + ak[l];
      tab[128 + l] = bk[l] * 8 + 4 * J + bcol[l], tab[192 + l] = (4 + bk[l]) * 8 + 4 * J + bcol[l];
      tab[256 + l] = (4 * (block[l] >> 1) + row[l]) * 8 + 4 * (block[l] & 1) + col[l];
    }
    printf("        A operand of lane l: block %d%d%d%d.. row/k e.g. lane 0:(%d,%d) 1:(%d,%d) 4:(%d,%d) 16:(%d,%d); B operand lane 0:(k %d,col %d) 1:(%d,%d) 4:(%d,%d)\n", ablock[0], ablock[16], ablock[32], ablock[48],
           arow[0], ak[0], arow[1], ak[1], arow[4], ak[4], arow[16], ak[16], bk[0], bcol[0], bk[1], bcol[1], bk[4], bcol[4]);
    hipMemcpy(dtab, tab.data(), 320 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k_mfma4, dim3(1), dim3(64), 0, 0, dP, dA, dout, dc, nrep, dtab);
      hipDeviceSynchronize();
      if (rep) report("mfma4", 1);
    }
  } else printf("4x4x4_4b layout not recognised (blocks %d)\n", nb);
  return 0;
}
