// Binding stub (test infrastructure, ours): exposes the reference's CPU K-nearest-neighbour routine
// `KNearestNeighborIdxCpu` (third_parties/pytorch3d/cuda/knn_cpu.cpp:13-69, compiled UNMODIFIED from
// /root/reference by oracle/build_ref.py) so that tests can pin oracle.knn / ia_voxelise_weights'
// neighbour sets against it.  The reference's own binding (knn.cpp) also pulls in the CUDA entry
// points, which need torch's hipify and a GPU; the CPU routine needs neither.
#include <torch/extension.h>
#include <tuple>

std::tuple<at::Tensor, at::Tensor> KNearestNeighborIdxCpu(const at::Tensor& p1, const at::Tensor& p2,
                                                          const at::Tensor& lengths1, const at::Tensor& lengths2,
                                                          const int norm, const int K);

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("knn_points_idx_cpu", &KNearestNeighborIdxCpu, "pytorch3d KNearestNeighborIdxCpu (reference source)");
}
