// Internal declarations shared by the host and device translation units of libbp5.
#pragma once
#include "../../include/bp5.h"

#include <string>
#include <vector>

namespace bp5 {

constexpr int MAXN = BP5_MAX_DEGREE + 1;

struct Tables {
  int n = 0;
  double nodes[MAXN], pts[MAXN], w[MAXN];
  double N[MAXN * MAXN], D[MAXN * MAXN]; // [q*n+i]
};

int shape_tables(int degree, int quadrature, Tables &t);
int fail(int status, const std::string &msg);

// Per-team index plan of the team-assembled operator kernel (bp5_kernels.hpp: apply_team_kernel):
// for each team of `cpt` consecutive cells the sorted distinct DoFs it touches (bit 31 set when no
// other team touches the DoF) and, per local DoF, its position in that list.
struct TeamPlanHost {
  std::vector<uint32_t> off, dofs, group_cell_off;
  std::vector<uint16_t> pos;
  // conflict-free accumulation rounds: cells of one team with equal round share no DoF
  std::vector<uint8_t> cell_round, team_rounds;
  // DoFs touched by more than one team: CSR of their partial-slab slots (slot = index into dofs),
  // slots ordered by team -> the combine kernel sums in a fixed order
  std::vector<uint32_t> sh_dof, sh_off, sh_slot;
  bool covers_all = false; // every local DoF is touched by some cell (overwrite mode is legal)
  // block kernel only: passes of `cells_per_pass` cell slots; entry = cell id, bit 31 set on an
  // idle slot (which then repeats a valid cell id so that loads stay in bounds)
  std::vector<uint32_t> pass_cell, pass_off;
};
int build_team_plan(const uint32_t *l2g, uint32_t n_cells, int n3, size_t n_local, int cpt, TeamPlanHost &out,
                    const uint32_t *blk_off = nullptr, uint32_t n_blocks = 0, int cells_per_pass = 0);

// z-marching plan (apply_march_kernel): cells are linked into chains along their local z direction
// (the k = n-1 face of a cell is the k = 0 face of its successor); `cpt` chains advance together.
// entry(team, step, slot) = cell id | flags: bit 31 idle slot, bit 30 "linked to the previous step"
struct MarchPlanHost {
  std::vector<uint32_t> team_off; // [n_teams+1] offsets (in steps) into entries / cpt
  std::vector<uint32_t> entries;  // [total_steps * cpt]
};
int build_march_plan(const uint32_t *l2g, uint32_t n_cells, int n, int cpt, int max_steps, MarchPlanHost &out);

} // namespace bp5
