// Operation-counting build of the oracle: the very source of rt_oracle.c compiled with its
// `double` replaced by a type that counts every add, subtract, multiply, divide, square root and
// pow it performs.  TEST INFRASTRUCTURE ONLY: it exists to check the algorithmic-flop constants
// that bench.py prices a frame with (tools/flop_check.py).  Results are unchanged (the counted
// type wraps a double and forwards every operation).
//
// Build: g++ -O1 -fpermissive -w -ffp-contract=off -fPIC -shared -pthread oracle/opcount.cpp -o oracle/libopcount.so
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <setjmp.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread uint64_t tl_ops[6];  // add/sub, mul, div, sqrt, pow, neg/abs/min/max/floor/compare (not flops)

struct counted {
    double v;
    counted() : v(0.0) {}
    template <typename T> counted(T x) : v((double)x) {}
    template <typename T> explicit operator T() const { return (T)v; }
};
#define OPC_BIN(op, slot)                                                                              \
    static inline counted operator op(counted a, counted b) { ++tl_ops[slot]; return counted(a.v op b.v); } \
    template <typename T> static inline counted operator op(counted a, T b) { ++tl_ops[slot]; return counted(a.v op (double)b); } \
    template <typename T> static inline counted operator op(T a, counted b) { ++tl_ops[slot]; return counted((double)a op b.v); }
OPC_BIN(+, 0)
OPC_BIN(-, 0)
OPC_BIN(*, 1)
OPC_BIN(/, 2)
static inline counted operator-(counted a) { ++tl_ops[5]; return counted(-a.v); }
static inline counted operator+(counted a) { return a; }
static inline counted &operator+=(counted &a, counted b) { ++tl_ops[0]; a.v += b.v; return a; }
static inline counted &operator-=(counted &a, counted b) { ++tl_ops[0]; a.v -= b.v; return a; }
static inline counted &operator*=(counted &a, counted b) { ++tl_ops[1]; a.v *= b.v; return a; }
static inline counted &operator/=(counted &a, counted b) { ++tl_ops[2]; a.v /= b.v; return a; }
#define OPC_CMP(op)                                                                      \
    static inline bool operator op(counted a, counted b) { ++tl_ops[5]; return a.v op b.v; }  \
    template <typename T> static inline bool operator op(counted a, T b) { ++tl_ops[5]; return a.v op (double)b; } \
    template <typename T> static inline bool operator op(T a, counted b) { ++tl_ops[5]; return (double)a op b.v; }
OPC_CMP(<)
OPC_CMP(>)
OPC_CMP(<=)
OPC_CMP(>=)
OPC_CMP(==)
OPC_CMP(!=)
static inline counted sqrt(counted a) { ++tl_ops[3]; return counted(::sqrt(a.v)); }
static inline counted pow(counted a, counted b) { ++tl_ops[4]; return counted(::pow(a.v, b.v)); }
static inline counted sin(counted a) { return counted(::sin(a.v)); }  // scene set-up only
static inline counted cos(counted a) { return counted(::cos(a.v)); }
static inline counted tan(counted a) { return counted(::tan(a.v)); }
static inline counted fabs(counted a) { ++tl_ops[5]; return counted(::fabs(a.v)); }
static inline counted floor(counted a) { ++tl_ops[5]; return counted(::floor(a.v)); }
static inline counted fmin(counted a, counted b) { ++tl_ops[5]; return counted(::fmin(a.v, b.v)); }
static inline counted fmax(counted a, counted b) { ++tl_ops[5]; return counted(::fmax(a.v, b.v)); }
static inline bool opc_isnan(counted a) { return a.v != a.v; }
static inline bool opc_isfinite(counted a) { return ::isfinite(a.v); }
#undef isnan
#undef isfinite
#define isnan(x) opc_isnan(x)
#define isfinite(x) opc_isfinite(x)

#define double counted
#include "rt_oracle.c"
#undef double

extern "C" {
// counters of the calling thread since the last reset: add/sub, mul, div, sqrt, pow, other
void opc_read(uint64_t *out6, int reset)
{
    for (int k = 0; k < 6; ++k) out6[k] = tl_ops[k];
    if (reset) for (int k = 0; k < 6; ++k) tl_ops[k] = 0;
}
}
