/*
 * gq_heightmap.h - the rays of the HeightMap sensor for ONE env (sensors/heightmap.py of the reference: create_sensor_matrix :106-169,
 * raycast_sensor :66-104): rows x cols vertical rays from a grid in the robot's heading frame, origin z = centre z + 0.6 - 0.07, direction -z,
 * static geoms only (the floor plane z = 0, the world boxes, the height field).  One wavefront: lane = box first picks the boxes whose bounding
 * circle meets the grid's circle (two ballots), lane = cell (strided) then walks those few.  Called by heightmap_kernel (gq_heightmap: any
 * centre / yaw the caller passes) and by the step kernel's epilogue (gq_batch_set_heightmap: the grid follows the base, no launch of its own).
 */
#pragma once
#include "gq_step_kernel.h"

namespace gq {

/* Mref: the model through the address space the caller holds it in.  out: the env's [rows * cols][3] hit points. */
template <class Mref>
__device__ inline void heightmap_rays(const Mref& M, const double cx, const double cy_, const double cz, const float cyaw, const float syaw, const int rows, const int cols,
                                      const float dist_x, const float dist_y, float* out) {
  const int lane = lane_id(), cells = rows * cols;
  const int nbox = M.nbox;
  uint64_t cand[2] = {0, 0};
  {
    const float reach = sqrtf((0.5f * rows + 1.0f) * dist_x * (0.5f * rows + 1.0f) * dist_x + (0.5f * cols + 1.0f) * dist_y * (0.5f * cols + 1.0f) * dist_y);
    for (int half = 0; half < 2 && half * GQ_WAVE < nbox; half++) { /* wave-uniform */
      const int b = half * GQ_WAVE + lane;
      bool near = false;
      if (b < nbox) {
        const double ox = cx - (double)M.box[b].pos[0], oy = cy_ - (double)M.box[b].pos[1], rr = (double)M.box[b].rad + (double)reach;
        near = ox * ox + oy * oy <= rr * rr;
      }
      cand[half] = ballot(near);
    }
  }
  for (int cell0 = 0; cell0 < cells; cell0 += GQ_WAVE) { /* wave-uniform; lanes past the last cell mirror it and do not store */
    const int cell = cell0 + lane < cells ? cell0 + lane : cells - 1, i = cell / cols, j = cell % cols;
    const float c_rows = (rows % 2 == 0) ? 0.5f * rows : 0.5f * (rows - 1), c_cols = (cols % 2 == 0) ? 0.5f * cols : 0.5f * (cols - 1);
    const float off_r = (rows % 2 == 0) ? -0.5f * dist_x : 0.0f, off_c = (cols % 2 == 0) ? -0.5f * dist_y : 0.0f;
    const float gx = dist_x * (c_rows - (float)i) + off_r, gy = dist_y * (c_cols - (float)j) + off_c;
    /* offset in the world frame: R_W2H^T [gx, gy], R_W2H = [[c, s], [-s, c]] */
    const double px = cx + (double)(cyaw * gx - syaw * gy);
    const double py = cy_ + (double)(syaw * gx + cyaw * gy);
    const double pz = cz + 0.6 - 0.07;
    /* mj_ray along -z against the floor plane: distance = pz (ray starts above the floor), hit = origin - z * dist */
    double dist = pz > 0.0 ? pz : -1.0;   /* mj_ray returns -1 when nothing is hit */
    for (int half = 0; half < 2; half++)
      for (uint64_t todo = cand[half]; todo; todo &= todo - 1) { /* wave-uniform */
        const int b = half * GQ_WAVE + ffs64(todo);
        const double ox = px - (double)M.box[b].pos[0], oy = py - (double)M.box[b].pos[1], oz = pz - (double)M.box[b].pos[2];
        if (ox * ox + oy * oy > (double)(M.box[b].rad * M.box[b].rad)) continue; /* the vertical ray misses the bounding sphere */
        /* origin and direction (0, 0, -1) in the box frame */
        double tin = 0.0, tout = 1e30;
        bool hit = true;
        for (int k = 0; k < 3 && hit; k++) {
          const double ol = (double)M.box[b].mat[k] * ox + (double)M.box[b].mat[3 + k] * oy + (double)M.box[b].mat[6 + k] * oz, dl = -(double)M.box[b].mat[6 + k];
          const double s = (double)M.box[b].size[k];
          if (fabs(dl) < 1e-12) { hit = fabs(ol) <= s; continue; }
          double t0 = (-s - ol) / dl, t1 = (s - ol) / dl;
          if (t0 > t1) { const double tt = t0; t0 = t1; t1 = tt; }
          if (t0 > tin) tin = t0;
          if (t1 < tout) tout = t1;
          hit = tin <= tout;
        }
        if (hit && tout >= 0.0 && (dist < 0.0 || tin < dist)) dist = tin;
      }
    if (M.hf_nrow > 0) { /* height field: the vertical ray meets the triangle under (px, py) */
      const float x = (float)(px - (double)M.hf_pos[0]), y = (float)(py - (double)M.hf_pos[1]);
      const float fx = (x + M.hf_sx) * M.hf_inv_dx, fy = (y + M.hf_sy) * M.hf_inv_dy;
      if (fx >= 0.0f && fy >= 0.0f && fx <= (float)(M.hf_ncol - 1) && fy <= (float)(M.hf_nrow - 1)) {
        const int nc = M.hf_ncol, c = imin((int)fx, nc - 2), r = imin((int)fy, M.hf_nrow - 2);
        const float u = fx - (float)c, v = fy - (float)r;
        const GQ_GLOBAL float* H = (const GQ_GLOBAL float*)M.hf_data;
        const float h00 = H[r * nc + c], h10 = H[r * nc + c + 1], h01 = H[(r + 1) * nc + c], h11 = H[(r + 1) * nc + c + 1];
        const float h = u + v <= 1.0f ? h00 + u * (h10 - h00) + v * (h01 - h00) : h11 + (1.0f - u) * (h01 - h11) + (1.0f - v) * (h10 - h11);
        const double top = (double)M.hf_pos[2] + (double)h, t = pz - top;
        if (t >= 0.0 && (dist < 0.0 || t < dist)) dist = t;
      }
    }
    if (cell0 + lane < cells) {
      GQ_GLOBAL float* o = (GQ_GLOBAL float*)out + (size_t)cell * 3;
      o[0] = (float)px; o[1] = (float)py; o[2] = (float)(pz - dist);
    }
  }
}

}  // namespace gq
