// What ONE wavefront pays for ONE pair of the convex routine (csrc/gq_convex.h: GJK + EPA), alone on its SIMD - the regime of a launch's tail waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Igym_quadruped_amd/csrc -Iinclude -DGQ_CVX_STATS -o /tmp/convex_pair tools/ubench/convex_pair.hip && /tmp/convex_pair
// Two random polytopes (NV vertices on an ellipsoid: every vertex is a hull vertex), B placed `depth` into A along a random direction; every
// block (one wavefront) runs its own pair REP times; prints shader cycles per pair, EPA iterations, and cycles per support query.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
namespace gq { __device__ long long gq_cvx_cyc[8]; }
#include "gq_convex.h"

#define REP 8
__global__ void __launch_bounds__(64) k_pair(const float* vx, const float* vy, const float* vz, const float* shapes, float margin, float* res, long long* cyc) {
  __shared__ float shp[GQ_CVX_SHP_WORDS], poly[GQ_CVX_POLY_WORDS];
  const int lane = threadIdx.x;
  const float* S = shapes + blockIdx.x * 2 * GQ_CVX_SHAPE_WORDS;
  if (lane < 2 * GQ_CVX_SHAPE_WORDS) shp[lane] = S[lane];
  __syncthreads();
  bool hit = false;
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++) hit = gq::cvx_pair_wave((gq::LdsF)shp, (gq::LdsF)poly, (const GQ_MODEL float*)vx, (const GQ_MODEL float*)vy, (const GQ_MODEL float*)vz, margin);
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0) { cyc[blockIdx.x] = (t1 - t0) / REP; res[blockIdx.x * 4] = hit ? shp[40] : 1e30f; res[blockIdx.x * 4 + 1] = (float)((int*)shp)[47]; }
}
// the pieces: 64 support queries in a dependent chain (the direction of query k + 1 comes from the answer of query k), and 64 tetrahedron
// closest-point evaluations
__global__ void __launch_bounds__(64) k_pieces(const float* vx, const float* vy, const float* vz, const float* shapes, float* res, long long* cyc) {
  __shared__ float shp[GQ_CVX_SHP_WORDS], poly[GQ_CVX_POLY_WORDS];
  const int lane = threadIdx.x;
  if (lane < 2 * GQ_CVX_SHAPE_WORDS) shp[lane] = shapes[lane];
  __syncthreads();
  const gq::CvxCaps caps = gq::cvx_caps_fetch((gq::LdsCF)shp, (gq::LdsCF)(shp + 20), (const GQ_MODEL float*)vx, (const GQ_MODEL float*)vy, (const GQ_MODEL float*)vz);
  const gq::CvxRegs G = gq::cvx_regs((gq::LdsCF)shp, (gq::LdsCF)(shp + 20));
  gq::V3 d = gq::v3(0.3f, 0.2f, 1.0f);
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int k = 0; k < 64; k++) {
    const gq::CvxMink m = gq::cvx_minkowski(G, (const GQ_MODEL float*)vx, (const GQ_MODEL float*)vy, (const GQ_MODEL float*)vz, d, caps);
    d = gq::v3(m.w.y + 0.1f, m.w.z - 0.3f, m.w.x + 0.2f);
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[0] = (t1 - t0) / 64;
  for (int i = 0; i < 4; i++) { poly[4 * i] = 0.3f + 0.01f * i * d.x; poly[4 * i + 1] = -0.2f + (i == 1 ? 0.5f : 0.0f); poly[4 * i + 2] = 0.1f + (i == 2 ? 0.4f : 0.0f) + (i == 3 ? d.y : 0.0f); }
  __syncthreads();
  float lam[4]; gq::V3 v = d; float acc = 0.0f;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int k = 0; k < 64; k++) {
    gq::cvx_simplex((gq::LdsCF)poly, 4, lam, v);
    acc += v.x + lam[1];
    if (lane == 0) poly[12] = 0.3f + 1e-6f * acc;
    __syncthreads();
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) { cyc[1] = (t1 - t0) / 64; res[0] = acc + d.x; }
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int k = 0; k < 64; k++) {
    gq::cvx_simplex((gq::LdsCF)poly, 3, lam, v);
    acc += v.x + lam[1];
    if (lane == 0) poly[8] = 0.3f + 1e-6f * acc;
    __syncthreads();
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) { cyc[2] = (t1 - t0) / 64; res[1] = acc; }
}
int main() {
  const int NV = 192, NP = 64;
  std::mt19937 rng(1);
  std::normal_distribution<float> N01(0, 1);
  std::uniform_real_distribution<float> U(0, 1);
  std::vector<float> vx, vy, vz, sh;
  auto cloud = [&](float a, float b, float c) { int adr = (int)vx.size(); for (int i = 0; i < NV; i++) { float x = N01(rng), y = N01(rng), z = N01(rng), l = std::sqrt(x * x + y * y + z * z); vx.push_back(a * x / l); vy.push_back(b * y / l); vz.push_back(c * z / l); } return adr; };
  for (int p = 0; p < NP; p++) {
    int aA = cloud(0.03f, 0.03f, 0.12f), aB = cloud(0.03f, 0.04f, 0.10f);
    // B end to end with A along z, a little off the axis: overlapping by `depth` at the tips (p % 4 == 0: 5 mm apart instead)
    float hA = -1e9f, hB = -1e9f;
    for (int i = 0; i < NV; i++) { hA = std::fmax(hA, vz[aA + i]); hB = std::fmax(hB, -vz[aB + i]); }
    const float gap = (p % 4 == 0) ? 0.005f : -(0.001f + 0.01f * U(rng));
    const float u[3] = {0.004f * N01(rng) / (hA + hB + gap), 0.004f * N01(rng) / (hA + hB + gap), 1.0f};
    float S[2 * GQ_CVX_SHAPE_WORDS] = {0};
    for (int s = 0; s < 2; s++) { int* I = (int*)(S + 20 * s); I[0] = 0; I[1] = s ? aB : aA; I[2] = NV; I[3] = -1; S[20 * s + 4] = S[20 * s + 8] = S[20 * s + 12] = 1.0f; }
    for (int k = 0; k < 3; k++) S[20 + 13 + k] = u[k] * (hA + hB + gap);
    sh.insert(sh.end(), S, S + 40);
  }
  float *dvx, *dvy, *dvz, *dsh, *dres; long long* dcyc;
  hipMalloc(&dvx, vx.size() * 4); hipMalloc(&dvy, vx.size() * 4); hipMalloc(&dvz, vx.size() * 4); hipMalloc(&dsh, sh.size() * 4); hipMalloc(&dres, NP * 16); hipMalloc(&dcyc, NP * 8);
  hipMemcpy(dvx, vx.data(), vx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dvy, vy.data(), vx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dvz, vz.data(), vx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dsh, sh.data(), sh.size() * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_pair, dim3(NP), dim3(64), 0, 0, dvx, dvy, dvz, dsh, 0.01f, dres, dcyc);
  std::vector<float> res(NP * 4); std::vector<long long> cyc(NP);
  hipMemcpy(res.data(), dres, NP * 16, hipMemcpyDeviceToHost); hipMemcpy(cyc.data(), dcyc, NP * 8, hipMemcpyDeviceToHost);
  double cs = 0, is = 0; int nc = 0;
  for (int p = 0; p < NP; p++) {
    if (p < 12) printf("pair %2d dist %9.6f epa its %2.0f cycles %7lld\n", p, res[4 * p], res[4 * p + 1], cyc[p]);
    if (res[4 * p] < 0) { cs += cyc[p]; is += res[4 * p + 1]; nc++; }
  }
  printf("%d penetrating pairs (%d vertices per hull, every chunk scanned): mean %.0f cycles per pair, %.1f EPA iterations -> %.0f cycles per EPA iteration (GJK not separated out)\n", nc, NV, cs / nc, is / nc, cs / is);
  { long long c[8]; hipMemcpyFromSymbol(c, HIP_SYMBOL(gq::gq_cvx_cyc), sizeof c);
    printf("cycle shares over all pairs and repeats: GJK glue %lld, support queries %lld, simplex + reduce %lld, EPA select/dup %lld, EPA patch + rim %lld, EPA fan + planes %lld\n", c[0], c[1], c[2], c[3], c[4], c[5]); }
  hipLaunchKernelGGL(k_pieces, dim3(1), dim3(64), 0, 0, dvx, dvy, dvz, dsh, dres, dcyc);
  hipLaunchKernelGGL(k_pieces, dim3(1), dim3(64), 0, 0, dvx, dvy, dvz, dsh, dres, dcyc);
  hipMemcpy(cyc.data(), dcyc, 3 * 8, hipMemcpyDeviceToHost);
  printf("pieces, one wave alone: support query of A - B (3 chunks each) %lld cycles, tetrahedron step %lld, triangle step %lld\n", cyc[0], cyc[1], cyc[2]);
  return 0;
}
