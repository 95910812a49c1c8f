// linear_poly / quadratic_poly with the surface the prover and verifier use
// (reference: src/polynomial.h:20-46, src/polynomial.cpp:64-131).
#pragma once
#include "field.hpp"

class linear_poly;

// a*x^2 + b*x + c
class quadratic_poly {
public:
    F a, b, c;
    quadratic_poly() {}
    quadratic_poly(const F &aa, const F &bb, const F &cc) : a(aa), b(bb), c(cc) {}
    quadratic_poly operator+(const quadratic_poly &x) const { return quadratic_poly(a + x.a, b + x.b, c + x.c); }
    quadratic_poly operator*(const F &x) const { return quadratic_poly(a * x, b * x, c * x); }
    F eval(const F &x) const { return (a * x + b) * x + c; }
};

// a*x + b
class linear_poly {
public:
    F a, b;
    linear_poly() {}
    linear_poly(const F &aa, const F &bb) : a(aa), b(bb) {}
    linear_poly(const F &x) : a(F_ZERO), b(x) {}
    linear_poly operator+(const linear_poly &x) const { return linear_poly(a + x.a, b + x.b); }
    quadratic_poly operator*(const linear_poly &x) const { return quadratic_poly(a * x.a, a * x.b + b * x.a, b * x.b); }
    linear_poly operator*(const F &x) const { return linear_poly(a * x, b * x); }
    F eval(const F &x) const { return a * x + b; }
};
