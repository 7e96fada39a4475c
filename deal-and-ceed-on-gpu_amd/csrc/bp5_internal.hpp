// Internal declarations shared by the host and device translation units of libbp5.
#pragma once
#include "../../include/bp5.h"

#include <string>

namespace bp5 {

constexpr int MAXN = BP5_MAX_DEGREE + 1;

struct Tables {
  int n = 0;
  double nodes[MAXN], pts[MAXN], w[MAXN];
  double N[MAXN * MAXN], D[MAXN * MAXN]; // [q*n+i]
};

int shape_tables(int degree, int quadrature, Tables &t);
int fail(int status, const std::string &msg);

} // namespace bp5
