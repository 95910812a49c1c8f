// Host-side field element with the surface of virgo::fieldElement that the prover/verifier/loader use
// (lib/virgo/src/fieldElement.hpp:19-105 of the reference): + - * == !=, inv, fastPow, getRootOfUnity,
// random, zero, one.  Arithmetic is the shared vp_field.h code (same functions the kernels run).
#pragma once
#include <cstdlib>
#include <vector>

#include "../csrc/vp_field.h"

namespace vph {

class fieldElement {
public:
    unsigned long long real, img;     // same member names/layout as the reference (fieldElement.hpp:96-97)

    fieldElement() : real(0), img(0) {}
    fieldElement(long long x) : real(x >= 0 ? (unsigned long long) x : vp::P61 + x), img(0) {}
    fieldElement(long long x, long long y)
        : real(x >= 0 ? (unsigned long long) x : vp::P61 + x), img(y >= 0 ? (unsigned long long) y : vp::P61 + y) {}

    static fieldElement from(const vp::F &f) { fieldElement r; r.real = f.re; r.img = f.im; return r; }
    vp::F raw() const { return vp::f_make(real, img); }

    fieldElement operator+(const fieldElement &o) const { return from(vp::f_add(raw(), o.raw())); }
    fieldElement operator-(const fieldElement &o) const { return from(vp::f_sub(raw(), o.raw())); }
    fieldElement operator-() const { return from(vp::f_neg(raw())); }
    fieldElement operator*(const fieldElement &o) const { return from(vp::f_mul(raw(), o.raw())); }
    bool operator==(const fieldElement &o) const { return real == o.real && img == o.img; }
    bool operator!=(const fieldElement &o) const { return !(*this == o); }
    fieldElement &operator+=(const fieldElement &o) { *this = *this + o; return *this; }
    fieldElement &operator-=(const fieldElement &o) { *this = *this - o; return *this; }
    fieldElement &operator*=(const fieldElement &o) { *this = *this * o; return *this; }

    static fieldElement zero() { return fieldElement(0ll); }
    static fieldElement one() { return fieldElement(1ll); }

    static fieldElement fastPow(fieldElement x, unsigned __int128 p) {
        fieldElement ret(1ll), t = x;
        while (p) { if (p & 1) ret = ret * t; t = t * t; p >>= 1; }
        return ret;
    }
    fieldElement inv() const { return fastPow(*this, (unsigned __int128) vp::P61 * vp::P61 - 2); }
    // order-2^log_order root: the reference's order-2^62 generator squared down (fieldElement.cpp:237-249)
    static fieldElement getRootOfUnity(int log_order) {
        fieldElement rou;
        rou.real = 2147483648ull; rou.img = 1033321771269002680ull;
        for (int i = 0; i < 62 - log_order; ++i) rou = rou * rou;
        return rou;
    }
    // verifier randomness: glibc random(), 2 x 20 decimal digits per element (fieldElement.cpp:119-124,362-367)
    static unsigned long long randomNumber() {
        unsigned long long ret = ::random() % 10;
        for (int i = 1; i < 20; ++i) ret = (ret * 10ull + (unsigned long long) (::random() % 10)) % vp::P61;
        return ret;
    }
    static fieldElement random() {
        fieldElement r;
        r.real = randomNumber() % vp::P61;
        r.img = randomNumber() % vp::P61;
        return r;
    }
    static void init() { srand(3396); }    // fieldElement.cpp:106-111
    static const unsigned long long mod = vp::P61;
};

}  // namespace vph

typedef vph::fieldElement F;
#define F_ONE (vph::fieldElement::one())
#define F_ZERO (vph::fieldElement::zero())
