/* TEST INFRASTRUCTURE ONLY -- C entry point around the reference's OWN marching cubes (class MCGpu and its kernels,
 * /root/reference/MCGpu/CudaKernels.cu:304-660), compiled for the host through shim/cuda.h.  The sequence of calls is the one
 * MCGpu.cpp:41-54 makes (init -> MC -> scaleVertices -> copy out).  Besides vertices and faces it returns, per vertex, the
 * lattice-edge key ((i*NY + j)*NZ + k)*3 + dir read back from the reference's d_edge_point_state_ table, which is what the
 * tests sort by (the reference's vertex order is whatever its atomics produce).  Never shipped. */
#define private public            /* the edge table is a private member of the reference class */
#include SR_REF_KERNELS           /* oracle/_ref/mc_ref_kernels.cpp: CudaKernels.cu with the launches rewritten */
#undef private
#include <cstdint>

extern "C" int mc_ref_run(const float* sdf, int NX, int NY, int NZ, float iso, float xs, float ys, float zs, float x0, float y0, float z0,
                          float* verts, int64_t* vkeys, int64_t cap_v, int64_t* faces, int64_t cap_f, int64_t* nv, int64_t* nf) {
  MCGpu& mc = MCGpu::Get(0);
  if (!mc.init(NX, NY, NZ)) return -1;
  /* the reference sizes its scratch at 5 % of the cells and never checks it (CudaKernels.cu:590-592): the caller (oracle/mc.py)
     counts what the volume needs with the C restatement first and refuses volumes that would overrun it */
  mc.MC(const_cast<float*>(sdf), iso);
  mc.scaleVertices(xs, ys, zs, x0, y0, z0);
  const int64_t V = mc.number_record_[0], F = mc.number_record_[1];
  *nv = V; *nf = F;
  if (V > cap_v || F > cap_f) return -3;
  memcpy(verts, mc.d_points_coor_, sizeof(float) * 3 * V);
  for (int64_t f = 0; f < 3 * F; ++f) faces[f] = (int64_t)mc.d_faces_index_[f];
  const int64_t nedge = (int64_t)NX * NY * NZ * 3;
  for (int64_t v = 0; v < V; ++v) vkeys[v] = -1;
  for (int64_t e = 0; e < nedge; ++e) {
    const int id = mc.d_edge_point_state_[e];
    if (id >= 0) vkeys[id] = e;
  }
  return 0;
}

/* Upper bound check the caller runs first: the reference's scratch capacities for this volume (CudaKernels.cu:590-592). */
extern "C" void mc_ref_capacity(int NX, int NY, int NZ, int64_t* cap_v, int64_t* cap_f) {
  *cap_v = (int)(NX * NY * NZ * 12 * 0.05);
  *cap_f = (int)(NX * NY * NZ * 5 * 0.05);
}
