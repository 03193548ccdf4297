"""Cycle model of Winograd F(2x2, 3x3) in the f16x2 arithmetic against the direct form, per 3x3 conv and 1005 windows.
direct:   3 matrix products per multiply-add (wh xl, wl xh, wh xh), v_mfma_f32_32x32x16_f16: 16 384 multiply-adds in 32 cycles of one SIMD.
winograd: 16 element-wise products per 2x2 output tile and channel pair instead of 36 (x 3 f16 products), plus the input transform on the
          VALU: per (tile, input channel) 16 values rebuilt from their halves (32 converts + 16 adds), B^T d B (32 adds), split again
          (8 + 8 packed converts, 16 converts back, 16 subtracts) = 128 vector lane-operations, repeated by every 32-output-channel block
          that stages the patch (the kernels run one 32-channel tile per block: tools/isa_classes.py); a wave
          instruction covers 64 lanes in 2 cycles of its SIMD at best (MI355X_MICROARCH.md: v_fma_f32 2 cycles, one wave alone 4).
Output transform (24 adds per tile and output channel, once per tile) is left out.  1024 SIMDs, 1.9 GHz under load.
measured: the launch that holds the conv today (profiles/r03_*; B launches include projection / pool / flatten work)."""
CLK = 1.9e9
SIMDS = 1024
W = 1005
layers = [  # name, cin, cout, H, W, measured us per 1005 windows (launch holding it)
    ("conv1_1.conv2", 32, 32, 128, 256, 2327), ("conv2_1.conv1", 32, 64, 64, 128, 972), ("conv2_1.conv2", 64, 64, 64, 128, 2490),
    ("conv3_1.conv1", 64, 96, 32, 64, 842), ("conv3_1.conv2", 96, 96, 32, 64, 1264), ("conv4_1.conv1", 96, 128, 16, 32, 389),
    ("conv4_1.conv2", 128, 128, 16, 32, 521), ("conv6.conv1", 256, 96, 16, 32, 672), ("conv6.conv2", 96, 96, 16, 32, 310),
    ("conv7.conv1", 192, 64, 32, 64, 1369), ("conv7.conv2", 64, 64, 32, 64, 566), ("conv8.conv1", 128, 32, 64, 128, 1906),
    ("conv8.conv2", 32, 32, 64, 128, 596), ("conv9_1.conv1", 64, 32, 128, 256, 3433), ("conv9_1.conv2", 32, 32, 128, 256, 2710)]
print("%-15s %9s | %8s %8s | %8s %8s %8s | %8s" % ("conv", "MMAC/win", "mfma us", "", "mfma us", "valu us", "sum us", "measured"))
print("%-15s %9s | %8s %8s | %8s %8s %8s | %8s" % ("", "", "direct", "", "winograd", "transform", "", "us"))
tot = [0, 0, 0, 0]
for name, ci, co, H, Wd, meas in layers:
    mac = ci * co * 9 * H * Wd
    t_direct = mac * 3 / 16384 * 32 / SIMDS / CLK * W * 1e6
    t_wino = t_direct / 2.25
    tile_ch = (H * Wd / 4) * ci * (co / 32)
    t_valu = tile_ch * 128 / 64 * 2 / SIMDS / CLK * W * 1e6
    tot[0] += t_direct; tot[1] += t_wino; tot[2] += t_valu; tot[3] += meas
    print("%-15s %9.1f | %8.0f %8s | %8.0f %8.0f %8.0f | %8d" % (name, mac / 1e6, t_direct, "", t_wino, t_valu, t_wino + t_valu, meas))
print("%-15s %9s | %8.0f %8s | %8.0f %8.0f %8.0f | %8d" % ("sum", "", tot[0], "", tot[1], tot[2], tot[1] + tot[2], tot[3]))
