"""What would block scales buy the fp8 mode?    CPU simulation, no GPU, no product code.

y = x W^T with x [rows, K], W [N, K] quantised to OCP e4m3 under three scaling schemes, error measured against the fp32 product:
  row     : one fp32 scale per row of x / per output channel of W              (what aigv_set_precision(FP8_LLM) ships, oracle/fp8.py)
  mx32    : one power-of-two (E8M0) scale per 32 consecutive k, x and W         (what v_mfma_scale_f32_16x16x128_f8f6f4 applies in hardware)
  blk128  : one fp32 scale per 128 consecutive k of x, per 128x128 block of W   (software block scales: partial sums rescaled on the VALU)
on (a) the synthetic model's own activation statistics (unit-variance rows behind an RMSNorm, N(0, 0.02^2) weights) and (b) the same
with outlier channels (8 of K columns 30x larger - what trained decoder checkpoints show in the residual stream).
python scripts/fp8_block_scale_sim.py"""
import torch

torch.manual_seed(0)
E4M3_MAX = 448.0


def q_e4m3(x):
    return x.to(torch.float8_e4m3fn).float()


def quant_row(x):
    amax = x.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    s = amax / E4M3_MAX
    return q_e4m3(x / s), s


def quant_block(x, blk, pow2):
    r, k = x.shape
    xb = x.reshape(r, k // blk, blk)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    s = amax / E4M3_MAX
    if pow2:   # E8M0: the next power of two at or above amax / 448 (no overflow of the quantised block)
        s = torch.exp2(torch.ceil(torch.log2(s)))
    return (q_e4m3(xb / s) * s).reshape(r, k)   # dequantised values: the scaled MFMA / the rescaled partial sums compute exactly this product


def w_block128(w):
    n, k = w.shape
    wb = w.reshape(n // 128, 128, k // 128, 128)
    amax = wb.abs().amax(dim=(1, 3), keepdim=True).clamp_min(1e-30)
    s = amax / E4M3_MAX
    return (q_e4m3(wb / s) * s).reshape(n, k)


def rel(y, ref):
    return ((y - ref).norm() / ref.norm()).item()


def case(name, x, w):
    ref = x.double() @ w.double().t()
    xq, xs = quant_row(x); wq, ws = quant_row(w)
    y_row = ((xq.double() @ wq.double().t()) * xs.double()) * ws.double().t()
    y_mx = quant_block(x, 32, True).double() @ quant_block(w, 32, True).double().t()
    y_b128 = quant_block(x, 128, False).double() @ w_block128(w).double().t()
    y_bf = x.bfloat16().double() @ w.bfloat16().double().t()
    print(f"{name:46s} rel. error of y:  row {rel(y_row, ref):.4f}   mx32 {rel(y_mx, ref):.4f}   blk128 {rel(y_b128, ref):.4f}   (bf16 operands {rel(y_bf, ref):.5f})")


rows, K, N = 512, 4096, 1024
x = torch.randn(rows, K)
w = torch.randn(N, K) * 0.02
case("(a) synthetic statistics (gaussian rows)", x, w)
xo = x.clone(); xo[:, torch.randperm(K)[:8]] *= 30.0
case("(b) 8 outlier channels x30", xo, w)
xo2 = x.clone(); xo2[:, torch.randperm(K)[:8]] *= 300.0
case("(c) 8 outlier channels x300", xo2, w)
