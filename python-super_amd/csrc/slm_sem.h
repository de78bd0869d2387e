// slm_sem.h -- Semantic-SuPer pieces of the GraphFit path (slm_sem.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "slm_gf.h"

// per-slot device scratch of the semantic terms (grow-only)
struct SemScratch {
  uint8_t* flags = nullptr;    // (C,H,W) boundary flags
  int32_t* sel = nullptr;      // selected flat indices c*H*W + y*W + x
  int32_t* counts = nullptr;   // (SLM_MAX_CLASSES + 1): per-class counts, [last] = selected total
  void* tmp = nullptr;
  float2* edge_xy = nullptr;
  double2* morph_g = nullptr;
  size_t cap_flags = 0, cap_sel = 0, cap_tmp = 0, cap_edge = 0, cap_morph = 0;
};

// boundary pixels of every class of sem.img_seg -> sc.edge_xy, edge_off[0..C]; synchronises st
hipError_t sem_extract_edges(SemScratch& sc, const slm_gf_semantic& sem, int H, int W, int32_t* edge_off,
                             hipStream_t st);
hipError_t sem_ensure_morph(SemScratch& sc, int N);
void sem_free(SemScratch& sc);
// morphing term, pass 1: per-surfel d(loss_i)/d(x,y), loss sum, kept count, candidate flag
void launch_gf_morph(GfSlot* slots, int n_frames, int maxN, hipStream_t st);
