// ek_pw.h -- numpy's pairwise sum, the parts shared with ek_features.hip (ek_pam.hip)
#pragma once
#include "ek_common.h"

// chunk trees over the leaf sums part[2 g + 0 / 1] (chunk sums behind them in
// `part`), the chunks added left to right -> out2[0], out2[1]
void ek_launch_pw_chunks_total(double *part, const EkPwShape *shapes, int n_full,
                               int n_leaves_total, int n_chunks, double *out2,
                               hipStream_t s);
