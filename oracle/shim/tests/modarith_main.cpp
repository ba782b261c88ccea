// TEST INFRASTRUCTURE ONLY. Translation unit for the reference's ModArith test suite, the part of its field tests
// that covers this backend's hot path (vector ops next to the NTT, Montgomery conversion, bit reverse and
// TYPED_TEST(ModArithTest, ntt), icicle/tests/test_mod_arithmetic_api.h:55-201,395-466,614-695). The reference
// includes that header from test_field_api.cpp, which also pulls in sumcheck / FRI / hash tests whose libraries are
// outside this backend's scope; here the header is compiled unmodified on its own. Which tests run on the HIP device is
// chosen with --gtest_filter by tests/test_gpu_reference_suite.py.
#include "test_mod_arithmetic_api.h"

int main(int argc, char** argv)
{
  ::testing::InitGoogleTest(&argc, argv);
  return RUN_ALL_TESTS();
}
