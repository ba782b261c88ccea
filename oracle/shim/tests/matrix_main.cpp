// TEST INFRASTRUCTURE ONLY. Translation unit for the reference's Matrix test suite (icicle/tests/test_matrix_api.h), compiled
// unmodified on its own; the reference includes that header from test_field_api.cpp together with sumcheck / FRI / hash tests
// whose libraries are outside this backend's scope. tests/test_gpu_reference_suite.py runs MatrixTest/*.matrixTranspose
// (test_matrix_api.h:450-505) with Main-device=HIP: the Rust NTT suite needs matrix_transpose on the main device
// (wrappers/rust/icicle-core/src/ntt/tests.rs:311-335). The matmul tests of the same header are not selected (no HIP matmul).
#include "test_matrix_api.h"

int main(int argc, char** argv)
{
  ::testing::InitGoogleTest(&argc, argv);
  return RUN_ALL_TESTS();
}
