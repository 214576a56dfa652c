// Host-side (CPU) pieces of the EER computation: pool-adjacent-violators and the ROC convex hull
// vertex walk of sidekit/bosaris/detplot.py:289-351,390-436.  These loops are inherently sequential
// and sort-bound (SURVEY K18 "on device or host"); they live in the library so the Python mirror
// does not interpret a million-element loop.  The floating-point operations and their order are
// those of the reference (compiled with -ffp-contract=off), so widths / heights / vertices are
// bit-identical to it.
#include <stdint.h>
#include <vector>

#include "../../include/sidekit_amd.h"

extern "C" {

int sk_pavx(const double* y, int64_t n, double* ghat_out, int64_t* width, double* height, int64_t* nbins) {
  if (!y || n <= 0 || !width || !height || !nbins) return SK_EARG;
  std::vector<double> g((size_t)n);
  std::vector<int64_t> len((size_t)n);
  int64_t ci = 0;
  g[0] = y[0];
  len[0] = 1;
  for (int64_t j = 1; j < n; ++j) {
    ++ci;
    g[ci] = y[j];
    len[ci] = 1;
    while (ci >= 1 && g[ci - 1] >= g[ci]) {  // pool adjacent violators (ties pool too)
      const int64_t nw = len[ci - 1] + len[ci];
      g[ci - 1] = g[ci - 1] + ((double)len[ci] / (double)nw) * (g[ci] - g[ci - 1]);
      len[ci - 1] = nw;
      --ci;
    }
  }
  *nbins = ci + 1;
  int64_t pos = 0;
  for (int64_t b = 0; b <= ci; ++b) {
    width[b] = len[b];
    height[b] = g[b];
    if (ghat_out)
      for (int64_t k = 0; k < len[b]; ++k) ghat_out[pos + k] = g[b];
    pos += len[b];
  }
  // reference quirk (detplot.py:343-349): the first bin's fill loop starts at j = 0 and also writes ghat[-1]
  if (ghat_out) ghat_out[n - 1] = g[0];
  return SK_OK;
}

// pideal: 1.0 for target / 0.0 for non-target, already ordered by ascending score (stable).
int sk_rocch_vertices(const double* pideal, int64_t n, int64_t n_tar, int64_t n_non, const int64_t* width, int64_t nbins,
                      double* pmiss, double* pfa) {
  if (!pideal || !width || !pmiss || !pfa || n != n_tar + n_non || nbins <= 0) return SK_EARG;
  int64_t left = 0;
  double miss = 0.0, fa = (double)n_non;
  double tar_left = 0.0;  // number of targets among the first `left` scores (exact: integer-valued)
  for (int64_t i = 0; i < nbins; ++i) {
    pmiss[i] = miss / (double)n_tar;
    pfa[i] = fa / (double)n_non;
    for (int64_t k = 0; k < width[i]; ++k) tar_left += pideal[left + k];
    left += width[i];
    miss = tar_left;
    fa = (double)(n - left) - ((double)n_tar - tar_left);
  }
  pmiss[nbins] = miss / (double)n_tar;
  pfa[nbins] = fa / (double)n_non;
  return SK_OK;
}

}  // extern "C"
