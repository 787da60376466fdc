"""Numeric gate of VERDICT r3 item 5, evaluated on the CPU BEFORE any kernel work: Winograd F(2x2, 3x3) with bf16 MFMA operands
(weight transform G g G^T in float64 rounded to bf16 once; input transform B^T d B add-only in fp32, rounded to bf16 as the MFMA
operand; fp32 accumulation; output transform in fp32) against the direct bf16 conv (bf16 weights, same bf16 inputs, fp32
accumulation), both measured against the fp32 conv of the same bf16-rounded inputs.  Acceptance: relative L2 error of the
Winograd path <= 1.25 x the direct kernel's.

    python tools/winograd_numeric_gate.py
"""
import torch
import torch.nn.functional as F

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def bf16(t):
    return t.to(torch.bfloat16).to(t.dtype)


def winograd_bf16(x, w, out_bf16=True):
    """x [N,C,H,W] (bf16-representable, fp32), w [K,C,3,3] fp32 (un-rounded folded weights) -> conv3x3 pad 1, fp32."""
    n, c, h, ww = x.shape
    k = w.shape[0]
    U = bf16(torch.einsum('ij,kcjl,ml->kcim', G, w.double(), G).float())           # [K,C,4,4] bf16 operand
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                       # [N,C,H/2,W/2,4,4]
    V = torch.einsum('ij,nchwjl,ml->nchwim', Bt.float(), tiles, Bt.float())          # exact in fp32 (sums of 4 bf16 values)
    V = bf16(V)                                                                      # MFMA operand rounding
    M = torch.einsum('kcim,nchwim->nkhwim', U.double(), V.double()).float()          # fp32-ish accumulation (products exact)
    Y = torch.einsum('ij,nkhwjl,ml->nkhwim', At.float(), M, At.float())              # [N,K,H/2,W/2,2,2]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, k, h, ww)
    return bf16(y) if out_bf16 else y


def main():
    torch.manual_seed(0)
    rows = []
    for name, c, k, relu_in in (('dec .3 512->512 (post-ReLU inputs)', 512, 512, True), ('dec .3 256->256', 256, 256, True),
                                ('lateral 64->256', 64, 256, True), ('signed inputs 256->256', 256, 256, False)):
        x = torch.randn(2, c, 32, 32)
        x = bf16(F.relu(x) if relu_in else x)
        w = torch.randn(k, c, 3, 3) / (c * 9) ** .5
        ref = F.conv2d(x.double(), w.double(), padding=1)
        direct = bf16(F.conv2d(x.double(), bf16(w).double(), padding=1).float())
        wino = winograd_bf16(x, w)
        wino_nr = winograd_bf16(x, w, out_bf16=False)
        direct_nr = F.conv2d(x.double(), bf16(w).double(), padding=1)
        e = lambda t: float((t.double() - ref).norm() / ref.norm())
        rows.append((name, e(direct), e(wino), e(wino) / e(direct), e(direct_nr), e(wino_nr)))
    print(f'{"layer":38s} {"direct":>10s} {"winograd":>10s} {"ratio":>7s}   (before the bf16 output rounding: direct, winograd)')
    for r in rows:
        print(f'{r[0]:38s} {r[1]:10.3e} {r[2]:10.3e} {r[3]:7.2f}   {r[4]:10.3e} {r[5]:10.3e}')
    worst = max(r[3] for r in rows)
    print(f'worst ratio {worst:.2f} -> gate (<= 1.25): {"PASS" if worst <= 1.25 else "FAIL"}')


if __name__ == '__main__':
    main()
