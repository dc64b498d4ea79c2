// internal interface of the row-block Winograd F(4,3) kernel (conv_wino.hip), dispatched by mrn_conv2d_x3_wino_hl32 (conv_x3.hip)
#pragma once

struct WinoRowsParams {
  const unsigned char* v;     // transformed activation [Gx][B][H][Wq][6][Cb][128 B]; Cb = Cin/32 HL32 lines, or (dense) Cin/64 fp16 lines
  const unsigned char* u;     // transformed weight [G][Cout][6][Cb][3][128 B]
  const float* bias;          // [G][N] or null
  const float* out_scale;     // [G][2] = {s, 1/s} of the weight prescale or null
  const float* x_scale;       // [2] = {s, 1/s} of the activation operand or null
  float* y;                   // [G][B][H][W][N]
  float* stats;               // [G][stats_blocks][2][N] or null
  long v_gstride, u_gstride;  // bytes (v_gstride 0: all groups read the same activation)
  long y_gstride;             // floats
  int v_bytes;                // bytes of one group's activation
  int G, B, H, W, Wq, Cb, N, act;
  int stats_blocks;           // row blocks the statistics buffer was sized for (mrn_conv2d_x3_wino_stats_floats)
  int dense;                  // 1 / 2: operands are plain fp16 / bfloat16, 64 channels per line (one product per term: the reduced-precision mode)
  int pool;                   // 1: y is [G][B][H/2][W/2][N], per 2x2 window the extreme chosen by the sign of the BatchNorm weight
  const long long* gamma;     // [G] device addresses of those BatchNorm weights, or null (all maxima)
  int tiles_p, row_blocks, tiles_n, tiles_m;   // filled by the launcher
};

bool mrn_wino_rows_supported(int H, int R, int Cout);
void mrn_wino_rows_select(int mode);
int mrn_launch_wino_rows(const WinoRowsParams& p, void* stream);
