// C-ABI door to the reference's own CPU nearest-neighbour search (utils/metrics/distance/cd/chamfer_distance.cpp:39-66,
// `nnsearch`), compiled from the reference's source where it lies (oracle/Makefile.ref).  Test infrastructure only:
// tests/test_oracle_golden.py pins oracle/metrics_oracle.py's Chamfer restatement to it.  Nothing of the reference is
// copied here - this file only declares the function and forwards to it.
void nnsearch(const int b, const int n, const int m, const float* xyz1, const float* xyz2, float* dist, int* idx);

extern "C" void ref_cd_nnsearch(int b, int n, int m, const float* xyz1, const float* xyz2, float* dist, int* idx) {
  nnsearch(b, n, m, xyz1, xyz2, dist, idx);
}
