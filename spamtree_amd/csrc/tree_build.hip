// Device parts of the tree builder (include/spamtree_tree.h).  Integer / comparison work bound by
// HBM and by atomics: no matrix cores here.  Every floating-point expression must round like the NumPy host path
// (spamtree_amd/topology.py), so contraction into FMAs is off for the whole file.
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>
#include <climits>

#include <cstdint>
#include <string>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "spamtree_hip.h"
#include "spamtree_tree.h"

extern "C" void st_set_create_error(const char *msg);   // spamtree_hip.hip: text behind st_last_error(NULL)

namespace {
struct Dev {   // a few device arrays with one-shot cleanup
  std::vector<void *> ptrs;
  ~Dev() { for (void *p : ptrs) (void)hipFree(p); }
  template <typename T> T *alloc(size_t n) {
    void *p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    return (T *)p;
  }
  template <typename T> T *upload(const T *src, size_t n) {
    T *d = alloc<T>(n);
    if (d && n && hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
  }
};
int fail(const std::string &m) { st_set_create_error(m.c_str()); return ST_ERR_HIP; }

__device__ __forceinline__ unsigned long long ordered_bits(double v) {   // monotone for v >= 0
  return (unsigned long long)__double_as_longlong(v);
}
__global__ void k_fill_u64(unsigned long long *a, long long n, unsigned long long v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}
__global__ void k_cell_minkey(const long long *code, const double *key, long long n, unsigned long long *mink) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) atomicMin(&mink[code[r]], ordered_bits(key[r]));
}
__global__ void k_cell_minix(const long long *code, const double *key, const long long *ix, long long n, const unsigned long long *mink,
                             unsigned long long *minix) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n && ordered_bits(key[r]) == mink[code[r]]) atomicMin(&minix[code[r]], (unsigned long long)ix[r]);
}
__global__ void k_cell_row(const long long *code, const double *key, const long long *ix, long long n, const unsigned long long *mink,
                           const unsigned long long *minix, long long *out_row) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n && ordered_bits(key[r]) == mink[code[r]] && (unsigned long long)ix[r] == minix[code[r]]) out_row[code[r]] = r;
}

// ---- nearest target of the same margin: targets binned into a uniform gx x gy grid (counting sort by cell through a radix
// sort of the cell ids), every query walks square rings of cells outwards until no unvisited cell can hold a closer target
struct Grid { double x0, y0, cw, ch; int gx, gy; };
__device__ __forceinline__ int cell_of(double v, double v0, double w, int g) {
  int c = (int)floor((v - v0) / w);
  return c < 0 ? 0 : (c >= g ? g - 1 : c);
}
__global__ void k_target_cells(const double *tx, const double *ty, long long nt, Grid G, unsigned int *cell, unsigned int *idx) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nt) { cell[t] = (unsigned int)(cell_of(ty[t], G.y0, G.ch, G.gy) * G.gx + cell_of(tx[t], G.x0, G.cw, G.gx)); idx[t] = (unsigned int)t; }
}
__global__ void k_cell_starts(const unsigned int *cell_sorted, long long nt, long long ncell, unsigned int *start) {
  // start[c] = first position p with cell_sorted[p] >= c (start[ncell] = nt): one binary search per cell
  const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c > ncell) return;
  long long lo = 0, hi = nt;
  while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((long long)cell_sorted[mid] < c) lo = mid + 1; else hi = mid; }
  start[c] = (unsigned int)lo;
}
__global__ void k_nearest(const double *tx, const double *ty, const int *tmv, const unsigned int *order, const unsigned int *start, Grid G,
                          const double *qx, const double *qy, const int *qmv, long long nq, const int *margin_has, long long *out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const double x = qx[i], y = qy[i];
  const int mv = qmv[i];
  const bool filter = margin_has[mv] != 0;
  const int cx = cell_of(x, G.x0, G.cw, G.gx), cy = cell_of(y, G.y0, G.ch, G.gy);
  // along an axis with ONE cell nothing is ever unvisited, so only the other axis bounds the distance to unvisited cells
  // (collinear targets -- a transect -- give such a grid: with min(cw, ch) ~ 1e-302 the early exit below never fired and
  // every query walked all 4096 rings)
  const double cmin = G.gy == 1 ? G.cw : (G.gx == 1 ? G.ch : (G.cw < G.ch ? G.cw : G.ch));
  double best = __builtin_inf();
  long long bi = -1;
  const int rmax = G.gx > G.gy ? G.gx : G.gy;
  for (int r = 0; r <= rmax; ++r) {
    if (bi >= 0) {   // every unvisited cell lies outside the (2r-1)-cell square: its targets are at least (r-1) cells away
      const double reach = (double)(r - 1) * cmin;
      if (reach > 0.0 && best < reach * reach) break;
    }
    const int y0 = cy - r, y1 = cy + r, x0 = cx - r, x1 = cx + r;
    for (int yy = y0; yy <= y1; ++yy) {
      if (yy < 0 || yy >= G.gy) continue;
      const bool edge_row = yy == y0 || yy == y1;
      for (int xx = x0; xx <= x1; xx += (edge_row ? 1 : (x1 - x0 > 0 ? x1 - x0 : 1))) {   // full top / bottom rows, the two end cells otherwise
        if (xx < 0 || xx >= G.gx) continue;
        const unsigned int c = (unsigned int)(yy * G.gx + xx);
        for (unsigned int p = start[c]; p < start[c + 1]; ++p) {
          const unsigned int t = order[p];
          if (filter && tmv[t] != mv) continue;
          const double dx = x - tx[t], dy = y - ty[t];
          const double d2 = dx * dx + dy * dy;
          if (d2 < best || (d2 == best && (long long)t < bi)) { best = d2; bi = (long long)t; }
        }
      }
    }
  }
  out[i] = bi;
}
}   // namespace

extern "C" int st_tb_sort(const double *x, int64_t n, int32_t device, double *sorted_out) {
  if (!x || !sorted_out || n < 0) return ST_ERR_USAGE;
  if (n > (int64_t)INT_MAX) { st_set_create_error("st_tb_sort: more than INT_MAX keys (hipCUB's item count is an int)"); return ST_ERR_USAGE; }
  if (n == 0) return ST_OK;
  if (hipSetDevice(device) != hipSuccess) return fail("no usable HIP device");
  Dev D;
  double *din = D.upload(x, (size_t)n), *dout = D.alloc<double>((size_t)n);
  if (!din || !dout) return fail("st_tb_sort: device allocation failed");
  size_t tmp = 0;
  if (hipcub::DeviceRadixSort::SortKeys(nullptr, tmp, din, dout, (int)n) != hipSuccess) return fail("st_tb_sort: radix sort (size query)");
  void *dtmp = D.alloc<char>(tmp);
  if (!dtmp) return fail("st_tb_sort: device allocation failed");
  if (hipcub::DeviceRadixSort::SortKeys(dtmp, tmp, din, dout, (int)n) != hipSuccess) return fail("st_tb_sort: radix sort");
  if (hipMemcpy(sorted_out, dout, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return fail("st_tb_sort: copy back");
  return ST_OK;
}

extern "C" int st_tb_cell_argmin(const int64_t *code, const double *key, const int64_t *ix, int64_t n, int64_t ncells, int32_t device,
                                 int64_t *out_row) {
  if (!code || !key || !ix || !out_row || n < 0 || ncells < 0) return ST_ERR_USAGE;
  for (int64_t c = 0; c < ncells; ++c) out_row[c] = -1;
  if (n == 0 || ncells == 0) return ST_OK;
  for (int64_t r = 0; r < n; ++r) if (code[r] < 0 || code[r] >= ncells || !(key[r] >= 0.0) || ix[r] < 0) { st_set_create_error("st_tb_cell_argmin: code / key / ix out of range"); return ST_ERR_USAGE; }
  if (hipSetDevice(device) != hipSuccess) return fail("no usable HIP device");
  Dev D;
  long long *dcode = (long long *)D.upload((const long long *)code, (size_t)n), *dix = (long long *)D.upload((const long long *)ix, (size_t)n);
  double *dkey = D.upload(key, (size_t)n);
  unsigned long long *mink = D.alloc<unsigned long long>((size_t)ncells), *minix = D.alloc<unsigned long long>((size_t)ncells);
  long long *drow = D.alloc<long long>((size_t)ncells);
  if (!dcode || !dix || !dkey || !mink || !minix || !drow) return fail("st_tb_cell_argmin: device allocation failed");
  const int T = 256;
  const unsigned gc = (unsigned)((ncells + T - 1) / T), gn = (unsigned)((n + T - 1) / T);
  hipLaunchKernelGGL(k_fill_u64, dim3(gc), dim3(T), 0, 0, mink, (long long)ncells, ~0ULL);
  hipLaunchKernelGGL(k_fill_u64, dim3(gc), dim3(T), 0, 0, minix, (long long)ncells, ~0ULL);
  hipLaunchKernelGGL(k_fill_u64, dim3(gc), dim3(T), 0, 0, (unsigned long long *)drow, (long long)ncells, ~0ULL);   // -1
  hipLaunchKernelGGL(k_cell_minkey, dim3(gn), dim3(T), 0, 0, dcode, dkey, (long long)n, mink);
  hipLaunchKernelGGL(k_cell_minix, dim3(gn), dim3(T), 0, 0, dcode, dkey, dix, (long long)n, mink, minix);
  hipLaunchKernelGGL(k_cell_row, dim3(gn), dim3(T), 0, 0, dcode, dkey, dix, (long long)n, mink, minix, drow);
  if (hipGetLastError() != hipSuccess) return fail("st_tb_cell_argmin: launch failed");
  if (hipMemcpy(out_row, drow, (size_t)ncells * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return fail("st_tb_cell_argmin: copy back");
  return ST_OK;
}

extern "C" int st_tb_nearest(const double *tx, const double *ty, const int32_t *tmv, int64_t nt, const double *qx, const double *qy,
                             const int32_t *qmv, int64_t nq, int32_t n_margins, int32_t device, int64_t *out_target) {
  if (!tx || !ty || !tmv || !qx || !qy || !qmv || !out_target || nt <= 0 || nq < 0 || n_margins < 1 || nt > 0x7fffffffLL) return ST_ERR_USAGE;
  if (nq == 0) return ST_OK;
  std::vector<int> has((size_t)n_margins, 0);
  double x0 = tx[0], x1 = tx[0], y0 = ty[0], y1 = ty[0];
  for (int64_t t = 0; t < nt; ++t) {
    if (tmv[t] < 0 || tmv[t] >= n_margins) { st_set_create_error("st_tb_nearest: target margin out of range"); return ST_ERR_USAGE; }
    has[tmv[t]] = 1;
    x0 = std::min(x0, tx[t]); x1 = std::max(x1, tx[t]); y0 = std::min(y0, ty[t]); y1 = std::max(y1, ty[t]);
  }
  for (int64_t i = 0; i < nq; ++i) if (qmv[i] < 0 || qmv[i] >= n_margins) { st_set_create_error("st_tb_nearest: query margin out of range"); return ST_ERR_USAGE; }
  if (hipSetDevice(device) != hipSuccess) return fail("no usable HIP device");
  // about two targets per cell, aspect ratio of the bounding box
  Grid G;
  const double wx = std::max(x1 - x0, 1e-300), wy = std::max(y1 - y0, 1e-300);
  const double cells = std::max(1.0, (double)nt / 2.0);
  const bool flat_y = (y1 - y0) <= 1e-12 * (x1 - x0), flat_x = (x1 - x0) <= 1e-12 * (y1 - y0);   // all targets on one line (or one point)
  if (flat_x && flat_y) { G.gx = 1; G.gy = 1; }
  else if (flat_y) { G.gx = (int)std::min(4096.0, std::max(1.0, std::floor(cells))); G.gy = 1; }
  else if (flat_x) { G.gy = (int)std::min(4096.0, std::max(1.0, std::floor(cells))); G.gx = 1; }
  else {
    G.gx = (int)std::min(4096.0, std::max(1.0, std::floor(std::sqrt(cells * wx / wy))));
    G.gy = (int)std::min(4096.0, std::max(1.0, std::floor(cells / G.gx)));
  }
  G.x0 = x0; G.y0 = y0; G.cw = wx / G.gx * (1.0 + 1e-12); G.ch = wy / G.gy * (1.0 + 1e-12);
  const long long ncell = (long long)G.gx * G.gy;
  Dev D;
  double *dtx = D.upload(tx, (size_t)nt), *dty = D.upload(ty, (size_t)nt), *dqx = D.upload(qx, (size_t)nq), *dqy = D.upload(qy, (size_t)nq);
  int *dtmv = D.upload((const int *)tmv, (size_t)nt), *dqmv = D.upload((const int *)qmv, (size_t)nq), *dhas = D.upload(has.data(), has.size());
  unsigned int *cell = D.alloc<unsigned int>((size_t)nt), *idx = D.alloc<unsigned int>((size_t)nt), *cell_s = D.alloc<unsigned int>((size_t)nt),
               *idx_s = D.alloc<unsigned int>((size_t)nt), *start = D.alloc<unsigned int>((size_t)ncell + 1);
  long long *dout = D.alloc<long long>((size_t)nq);
  if (!dtx || !dty || !dqx || !dqy || !dtmv || !dqmv || !dhas || !cell || !idx || !cell_s || !idx_s || !start || !dout) return fail("st_tb_nearest: device allocation failed");
  const int T = 256;
  hipLaunchKernelGGL(k_target_cells, dim3((unsigned)((nt + T - 1) / T)), dim3(T), 0, 0, dtx, dty, (long long)nt, G, cell, idx);
  size_t tmp = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, cell, cell_s, idx, idx_s, (int)nt) != hipSuccess) return fail("st_tb_nearest: radix sort (size query)");
  void *dtmp = D.alloc<char>(tmp);
  if (!dtmp) return fail("st_tb_nearest: device allocation failed");
  if (hipcub::DeviceRadixSort::SortPairs(dtmp, tmp, cell, cell_s, idx, idx_s, (int)nt) != hipSuccess) return fail("st_tb_nearest: radix sort");   // stable: targets of a cell stay in index order
  hipLaunchKernelGGL(k_cell_starts, dim3((unsigned)((ncell + 1 + T - 1) / T)), dim3(T), 0, 0, cell_s, (long long)nt, ncell, start);
  hipLaunchKernelGGL(k_nearest, dim3((unsigned)((nq + T - 1) / T)), dim3(T), 0, 0, dtx, dty, dtmv, idx_s, start, G, dqx, dqy, dqmv, (long long)nq, dhas, dout);
  if (hipGetLastError() != hipSuccess) return fail("st_tb_nearest: launch failed");
  if (hipMemcpy(out_target, dout, (size_t)nq * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return fail("st_tb_nearest: copy back");
  return ST_OK;
}
