// Minimal stand-in for GoogleTest (v1.13 is fetched from the network by the reference's CMake,
// icicle/tests/CMakeLists.txt:1-8, which is impossible here). TEST INFRASTRUCTURE ONLY: it lets the reference's own
// test sources (icicle/tests/test_curve_api.cpp, test_device_api.cpp, test_mod_arithmetic_api.h) compile UNMODIFIED
// and run against the HIP backend plugin (oracle/build_ref_tests.sh, tests/test_gpu_reference_suite.py).
// Implements exactly what those files use: TEST / TEST_F / TYPED_TEST_SUITE / TYPED_TEST, testing::Test and
// testing::Types, ASSERT_/EXPECT_ {EQ,NE,GT,GE,LT,LE,TRUE,FALSE}, EXPECT_ANY_THROW / EXPECT_NO_THROW, GTEST_SKIP,
// message streaming, InitGoogleTest with --gtest_filter / --gtest_list_tests, RUN_ALL_TESTS. Written from the
// public GoogleTest documentation; it carries no test logic.
#pragma once
#include <cstdio>
#include <cstring>
#include <exception>
#include <functional>
#include <iostream>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

namespace testing {

  class Test
  {
  public:
    virtual ~Test() = default;
    static void SetUpTestSuite() {}
    static void TearDownTestSuite() {}
    virtual void SetUp() {}
    virtual void TearDown() {}
    virtual void TestBody() = 0;
  };

  template <typename... Ts>
  struct Types {
  };

  class Message
  {
  public:
    Message() = default;
    Message(const Message& o) { ss_ << o.str(); }
    template <typename T>
    Message& operator<<(const T& v)
    {
      ss_ << v;
      return *this;
    }
    Message& operator<<(std::ostream& (*manip)(std::ostream&))
    {
      ss_ << manip;
      return *this;
    }
    std::string str() const { return ss_.str(); }

  private:
    std::ostringstream ss_;
  };

  namespace internal {
    struct TestInfo {
      std::string suite, name;
      std::function<Test*()> factory;
      void (*setup_suite)();
      void (*teardown_suite)();
    };
    inline std::vector<TestInfo>& registry()
    {
      static std::vector<TestInfo> r;
      return r;
    }
    struct State {
      bool failed = false, skipped = false;
    };
    inline State& state()
    {
      static thread_local State* cur = nullptr;
      static State global;
      (void)cur;
      return global;
    }
    inline std::string& filter()
    {
      static std::string f = "*";
      return f;
    }
    inline bool& list_only()
    {
      static bool b = false;
      return b;
    }

    // streams v if it is streamable, a placeholder otherwise
    template <typename T, typename = void>
    struct Printable : std::false_type {
    };
    template <typename T>
    struct Printable<T, std::void_t<decltype(std::declval<std::ostream&>() << std::declval<const T&>())>> : std::true_type {
    };
    template <typename T>
    std::string print(const T& v)
    {
      if constexpr (std::is_enum_v<T>) {
        return std::to_string((long long)v);
      } else if constexpr (Printable<T>::value) {
        std::ostringstream ss;
        ss << v;
        return ss.str();
      } else {
        return "<" + std::to_string(sizeof(T)) + "-byte object>";
      }
    }

    class AssertHelper
    {
    public:
      AssertHelper(const char* file, int line, std::string what, bool skip = false) : file_(file), line_(line), what_(std::move(what)), skip_(skip) {}
      void operator=(const Message& m) const
      {
        if (skip_) {
          state().skipped = true;
          std::printf("%s:%d: Skipped %s\n", file_, line_, m.str().c_str());
        } else {
          state().failed = true;
          std::printf("%s:%d: Failure\n%s\n%s\n", file_, line_, what_.c_str(), m.str().c_str());
        }
        std::fflush(stdout);
      }

    private:
      const char* file_;
      int line_;
      std::string what_;
      bool skip_;
    };

    template <typename A, typename B>
    std::string cmp_msg(const char* op, const char* ea, const char* eb, const A& a, const B& b)
    {
      return std::string("Expected: (") + ea + ") " + op + " (" + eb + "), actual: " + print(a) + " vs " + print(b);
    }

    inline bool wild(const char* p, const char* s)
    {
      if (*p == 0) return *s == 0;
      if (*p == '*') return wild(p + 1, s) || (*s && wild(p, s + 1));
      if (*s && (*p == '?' || *p == *s)) return wild(p + 1, s + 1);
      return false;
    }
    inline bool any_match(const std::string& pats, const std::string& name)
    {
      size_t i = 0;
      while (i <= pats.size()) {
        size_t j = pats.find(':', i);
        if (j == std::string::npos) j = pats.size();
        if (j > i && wild(pats.substr(i, j - i).c_str(), name.c_str())) return true;
        i = j + 1;
      }
      return false;
    }
    inline bool selected(const std::string& full)
    {
      const std::string& f = filter();
      const size_t dash = f.find('-');
      const std::string pos = dash == std::string::npos ? f : f.substr(0, dash);
      const std::string neg = dash == std::string::npos ? "" : f.substr(dash + 1);
      return (pos.empty() || any_match(pos, full)) && !(neg.size() && any_match(neg, full));
    }

    template <typename F>
    bool register_test(const char* suite, const char* name, void (*su)(), void (*td)())
    {
      registry().push_back({suite, name, []() -> Test* { return new F(); }, su, td});
      return true;
    }
    template <template <typename> class TT, typename List>
    struct TypedRegistrar;
    template <template <typename> class TT, typename... Ts>
    struct TypedRegistrar<TT, Types<Ts...>> {
      static bool run(const char* suite, const char* name)
      {
        int idx = 0;
        (void)std::initializer_list<int>{(registry().push_back({std::string(suite) + "/" + std::to_string(idx++), name, []() -> Test* { return new TT<Ts>(); },
                                                                &TT<Ts>::SetUpTestSuite, &TT<Ts>::TearDownTestSuite}),
                                          0)...};
        return true;
      }
    };
  } // namespace internal

  inline void InitGoogleTest(int* argc, char** argv)
  {
    for (int i = 1; i < *argc; i++) {
      if (!std::strncmp(argv[i], "--gtest_filter=", 15)) internal::filter() = argv[i] + 15;
      if (!std::strcmp(argv[i], "--gtest_list_tests")) internal::list_only() = true;
    }
  }
  inline void InitGoogleTest() {}

  inline int RunAllTests()
  {
    using namespace internal;
    int ran = 0, failed = 0, skipped = 0;
    std::vector<std::string> failures;
    std::string cur_suite;
    void (*cur_td)() = nullptr;
    for (auto& t : registry()) {
      const std::string full = t.suite + "." + t.name;
      if (!selected(full)) continue;
      if (list_only()) {
        std::printf("%s\n", full.c_str());
        continue;
      }
      bool suite_ok = true;
      if (t.suite != cur_suite) {
        if (cur_td) cur_td();
        cur_suite = t.suite;
        cur_td = t.teardown_suite;
        state() = State{};
        try {
          if (t.setup_suite) t.setup_suite();
        } catch (const std::exception& e) {
          std::printf("SetUpTestSuite threw: %s\n", e.what());
          state().failed = true;
        }
        suite_ok = !state().failed;
      }
      std::printf("[ RUN      ] %s\n", full.c_str());
      std::fflush(stdout);
      state() = State{};
      state().failed = !suite_ok;
      if (suite_ok) {
        Test* obj = nullptr;
        try {
          obj = t.factory();
          obj->SetUp();
          if (!state().failed && !state().skipped) obj->TestBody();
          obj->TearDown();
        } catch (const std::exception& e) {
          std::printf("unexpected exception: %s\n", e.what());
          state().failed = true;
        } catch (...) {
          std::printf("unexpected exception of unknown type\n");
          state().failed = true;
        }
        delete obj;
      }
      ran++;
      if (state().failed) {
        failed++;
        failures.push_back(full);
        std::printf("[  FAILED  ] %s\n", full.c_str());
      } else if (state().skipped) {
        skipped++;
        std::printf("[  SKIPPED ] %s\n", full.c_str());
      } else {
        std::printf("[       OK ] %s\n", full.c_str());
      }
      std::fflush(stdout);
    }
    if (cur_td) cur_td();
    if (list_only()) return 0;
    std::printf("[==========] %d tests ran.\n[  PASSED  ] %d tests.\n", ran, ran - failed - skipped);
    if (skipped) std::printf("[  SKIPPED ] %d tests.\n", skipped);
    if (failed) {
      std::printf("[  FAILED  ] %d tests, listed below:\n", failed);
      for (auto& f : failures)
        std::printf("[  FAILED  ] %s\n", f.c_str());
    }
    if (ran == 0) std::printf("no test matched the filter\n");
    return failed ? 1 : 0;
  }
} // namespace testing

#define RUN_ALL_TESTS() ::testing::RunAllTests()

#define GTEST_AMBIGUOUS_ELSE_BLOCKER_                                                                                  \
  switch (0)                                                                                                           \
  case 0:                                                                                                              \
  default:

#define GTEST_FATAL_(what) return ::testing::internal::AssertHelper(__FILE__, __LINE__, what) = ::testing::Message()
#define GTEST_NONFATAL_(what) ::testing::internal::AssertHelper(__FILE__, __LINE__, what) = ::testing::Message()
#define GTEST_SKIP() return ::testing::internal::AssertHelper(__FILE__, __LINE__, "", true) = ::testing::Message()

#define GTEST_CMP_(a, b, op, opname, on_fail)                                                                          \
  GTEST_AMBIGUOUS_ELSE_BLOCKER_                                                                                        \
  if (const auto& gtest_a_ = (a); true)                                                                                \
    if (const auto& gtest_b_ = (b); gtest_a_ op gtest_b_)                                                              \
      ;                                                                                                                \
    else                                                                                                               \
      on_fail(::testing::internal::cmp_msg(opname, #a, #b, gtest_a_, gtest_b_))

#define ASSERT_EQ(a, b) GTEST_CMP_(a, b, ==, "==", GTEST_FATAL_)
#define ASSERT_NE(a, b) GTEST_CMP_(a, b, !=, "!=", GTEST_FATAL_)
#define ASSERT_GT(a, b) GTEST_CMP_(a, b, >, ">", GTEST_FATAL_)
#define ASSERT_GE(a, b) GTEST_CMP_(a, b, >=, ">=", GTEST_FATAL_)
#define ASSERT_LT(a, b) GTEST_CMP_(a, b, <, "<", GTEST_FATAL_)
#define ASSERT_LE(a, b) GTEST_CMP_(a, b, <=, "<=", GTEST_FATAL_)
#define EXPECT_EQ(a, b) GTEST_CMP_(a, b, ==, "==", GTEST_NONFATAL_)
#define EXPECT_NE(a, b) GTEST_CMP_(a, b, !=, "!=", GTEST_NONFATAL_)
#define EXPECT_GT(a, b) GTEST_CMP_(a, b, >, ">", GTEST_NONFATAL_)
#define EXPECT_GE(a, b) GTEST_CMP_(a, b, >=, ">=", GTEST_NONFATAL_)
#define EXPECT_LT(a, b) GTEST_CMP_(a, b, <, "<", GTEST_NONFATAL_)
#define EXPECT_LE(a, b) GTEST_CMP_(a, b, <=, "<=", GTEST_NONFATAL_)

#define GTEST_BOOL_(c, expected, on_fail)                                                                              \
  GTEST_AMBIGUOUS_ELSE_BLOCKER_                                                                                        \
  if (static_cast<bool>(c) == expected)                                                                                \
    ;                                                                                                                  \
  else                                                                                                                 \
    on_fail(std::string("Value of: " #c "\n  Actual: ") + (expected ? "false" : "true") + "\nExpected: " + (expected ? "true" : "false"))
#define ASSERT_TRUE(c) GTEST_BOOL_(c, true, GTEST_FATAL_)
#define ASSERT_FALSE(c) GTEST_BOOL_(c, false, GTEST_FATAL_)
#define EXPECT_TRUE(c) GTEST_BOOL_(c, true, GTEST_NONFATAL_)
#define EXPECT_FALSE(c) GTEST_BOOL_(c, false, GTEST_NONFATAL_)

#define GTEST_THROW_(stmt, want_throw, on_fail)                                                                        \
  GTEST_AMBIGUOUS_ELSE_BLOCKER_                                                                                        \
  if (bool gtest_threw_ = false; true) {                                                                               \
    try {                                                                                                              \
      stmt;                                                                                                            \
    } catch (...) {                                                                                                    \
      gtest_threw_ = true;                                                                                             \
    }                                                                                                                  \
    if (gtest_threw_ != want_throw) on_fail(std::string("Expected: " #stmt) + (want_throw ? " throws an exception.\n  Actual: it doesn't." : " doesn't throw an exception.\n  Actual: it throws.")); \
  } else                                                                                                               \
    (void)0
#define EXPECT_ANY_THROW(stmt) GTEST_THROW_(stmt, true, GTEST_NONFATAL_)
#define EXPECT_NO_THROW(stmt) GTEST_THROW_(stmt, false, GTEST_NONFATAL_)
#define ASSERT_ANY_THROW(stmt) GTEST_THROW_(stmt, true, GTEST_FATAL_)
#define ASSERT_NO_THROW(stmt) GTEST_THROW_(stmt, false, GTEST_FATAL_)

#define GTEST_TEST_CLASS_(suite, name) suite##_##name##_Test

#define GTEST_DEFINE_TEST_(suite, name, parent)                                                                        \
  class GTEST_TEST_CLASS_(suite, name) : public parent                                                                 \
  {                                                                                                                    \
  public:                                                                                                              \
    void TestBody() override;                                                                                          \
  };                                                                                                                   \
  static bool gtest_reg_##suite##_##name##_ =                                                                          \
    ::testing::internal::register_test<GTEST_TEST_CLASS_(suite, name)>(#suite, #name, &parent::SetUpTestSuite, &parent::TearDownTestSuite); \
  void GTEST_TEST_CLASS_(suite, name)::TestBody()

#define TEST(suite, name) GTEST_DEFINE_TEST_(suite, name, ::testing::Test)
#define TEST_F(fixture, name) GTEST_DEFINE_TEST_(fixture, name, fixture)

#define TYPED_TEST_SUITE(suite, types, ...) typedef types gtest_type_params_##suite##_
#define TYPED_TEST(suite, name)                                                                                        \
  template <typename gtest_TypeParam_>                                                                                 \
  class GTEST_TEST_CLASS_(suite, name) : public suite<gtest_TypeParam_>                                                \
  {                                                                                                                    \
  public:                                                                                                              \
    typedef suite<gtest_TypeParam_> TestFixture;                                                                       \
    typedef gtest_TypeParam_ TypeParam;                                                                                \
    void TestBody() override;                                                                                          \
  };                                                                                                                   \
  static bool gtest_reg_##suite##_##name##_ =                                                                          \
    ::testing::internal::TypedRegistrar<GTEST_TEST_CLASS_(suite, name), gtest_type_params_##suite##_>::run(#suite, #name); \
  template <typename gtest_TypeParam_>                                                                                 \
  void GTEST_TEST_CLASS_(suite, name)<gtest_TypeParam_>::TestBody()
