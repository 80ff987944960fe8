// capi.hip - ABI version of libtamtr_hip.so (the kernels live in gate/msdeform/contrastive/selfattn/selscan/gemm_bf16/lsap/cpam/dwconv/ss2d_out/bn/conv3x3/imgaug/fold/optim/xproj/detrloss.hip).
#include "common.h"
#include <stdlib.h>

extern "C" int tamtr_abi_version(void) { return 34; }


// Node census of the graph a stream is currently capturing into: counts[t] = number of nodes of hipGraphNodeType t (t < n_types <= 16).
// graphs.GraphedPart calls it at the end of each capture: a recorded part that holds MEMSET nodes (type 2) does not replay correctly
// under the runtime's AQL packet capture (profiles/r04_packet_capture_bisect.txt), and which library solver or torch op slips one in
// depends on shapes and tables - so it is counted, not assumed.  TAMTR_EINVAL when the stream is not capturing.
extern "C" int tamtr_graph_capture_census(void* stream, int* counts, int n_types) {
  if (!counts || n_types <= 0 || n_types > 16) return TAMTR_EINVAL;
  for (int i = 0; i < n_types; ++i) counts[i] = 0;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  hipGraph_t graph = nullptr;
  if (hipStreamGetCaptureInfo_v2((hipStream_t)stream, &st, nullptr, &graph, nullptr, nullptr) != hipSuccess) return TAMTR_ELAUNCH;
  if (st != hipStreamCaptureStatusActive || !graph) return TAMTR_EINVAL;
  size_t n = 0;
  if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess) return TAMTR_ELAUNCH;
  if (n == 0) return TAMTR_OK;
  hipGraphNode_t* nodes = (hipGraphNode_t*)malloc(n * sizeof(hipGraphNode_t));
  if (!nodes) return TAMTR_ELAUNCH;
  int rc = TAMTR_OK;
  if (hipGraphGetNodes(graph, nodes, &n) != hipSuccess) rc = TAMTR_ELAUNCH;
  for (size_t i = 0; rc == TAMTR_OK && i < n; ++i) {
    hipGraphNodeType t;
    if (hipGraphNodeGetType(nodes[i], &t) != hipSuccess) { rc = TAMTR_ELAUNCH; break; }
    if ((int)t >= 0 && (int)t < n_types) counts[(int)t]++;
  }
  free(nodes);
  return rc;
}
