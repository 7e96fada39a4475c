/* Pure-C consumer of include/bp5.h: the header must compile as C11 and the library must link from C.
 * Exercises the host-only entry points (no GPU needed) and the error path of a compute entry point. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "bp5.h"

#define CHECK(cond)                                                  \
  do {                                                               \
    if (!(cond)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

int main(void)
{
  double nodes[9], pts[9], w[9], N[81], D[81];
  CHECK(bp5_shape_tables(4, BP5_QUAD_GAUSS, nodes, pts, w, N, D) == BP5_OK);
  double ws = 0, rs = 0;
  for (int i = 0; i < 5; ++i) { ws += w[i]; rs += N[2 * 5 + i]; }
  CHECK(fabs(ws - 1.0) < 1e-14 && fabs(rs - 1.0) < 1e-13);
  CHECK(bp5_shape_tables(0, BP5_QUAD_GAUSS, nodes, pts, w, N, D) == BP5_ERR_INVALID);
  CHECK(strlen(bp5_last_error()) > 0 && strcmp(bp5_strerror(BP5_ERR_INVALID), "invalid argument") == 0);

  bp5_mesh_desc md;
  memset(&md, 0, sizeof md);
  md.degree = 2; md.cells[0] = 3; md.cells[1] = 2; md.cells[2] = 4; md.h = 0.5; md.rank = 1; md.n_ranks = 2;
  bp5_mesh *mesh = NULL;
  CHECK(bp5_mesh_create_brick(&md, &mesh) == BP5_OK);
  bp5_mesh_view v;
  CHECK(bp5_mesh_view_get(mesh, &v) == BP5_OK);
  CHECK(v.n_cells == 3 * 2 * 2 && v.n_ghost == 7 * 5 && v.n_neighbors == 1 && v.neighbor_rank_host[0] == 0);
  CHECK(v.n_interior_cells == 3 * 2 * 1);
  CHECK(v.n_global_dofs == 7u * 5u * 9u);
  for (uint32_t i = 0; i < v.n_cells * 27u; ++i) CHECK(v.local_to_global_host[i] < v.n_owned + v.n_ghost);

  /* compute entry points fail loudly without a device (or succeed with one) -- never a CPU fallback */
  int ndev = 0;
  int st = bp5_device_count(&ndev);
  bp5_mf_desc d;
  memset(&d, 0, sizeof d);
  d.dim = 3; d.degree = 2; d.quadrature = BP5_QUAD_GAUSS;
  d.n_cells = v.n_cells; d.n_interior_cells = v.n_interior_cells; d.n_owned = v.n_owned; d.n_ghost = v.n_ghost;
  d.local_to_global_host = v.local_to_global_host; d.node_coords_host = v.node_coords_host;
  d.constrained_host = v.constrained_host; d.n_constrained = v.n_constrained;
  d.n_neighbors = v.n_neighbors; d.neighbor_rank_host = v.neighbor_rank_host;
  d.send_offsets_host = v.send_offsets_host; d.send_indices_host = v.send_indices_host; d.recv_offsets_host = v.recv_offsets_host;
  bp5_mf *mf = NULL;
  st = bp5_mf_create(&d, &mf);
  if (ndev <= 0) CHECK(st == BP5_ERR_NO_DEVICE && mf == NULL);
  else { CHECK(st == BP5_OK); CHECK(bp5_mf_destroy(mf) == BP5_OK); }
  bp5_mesh_destroy(mesh);
  printf("abi_smoke ok (devices: %d)\n", ndev);
  return 0;
}
