// C shim (ours) around the REFERENCE's HungarianAlgorithm::assignmentoptimal (skeleton_3d/src/Hungarian.cpp:60),
// linked against the reference source compiled in place by oracle/Makefile.  Test infrastructure only: used to
// generate tests/golden/hungarian_golden.json and to cross-check the oracle's restatement.
#include <Hungarian.h>

extern "C" void hungarian_ref(int *assignment, double *cost, double *distMatrix, int nOfRows, int nOfColumns) {
  HungarianAlgorithm::assignmentoptimal(assignment, cost, distMatrix, nOfRows, nOfColumns);
}
