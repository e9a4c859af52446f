"""CPU restatement of the associative-embedding training loss and its gradient (numpy, float32).

TEST INFRASTRUCTURE ONLY (checker for the HIP loss kernels).
Parity status: PINNED by tests/golden/loss.npz (tools/make_golden.py imports the reference's AEKeypointsLoss in the
build container, evaluates it in fp32 on this repo's seeded synthetic batches and stores the losses together with
torch-autograd gradients of `hm0 + hm1 + push + pull` -- the sum of keypoints/module.py:50-59).

Follows /root/reference/src/keypoints/loss.py: HeatmapsLoss.forward :12-16, AEGroupingLoss.forward :20-61,
AEKeypointsLoss.calculate_loss :71-93 (push and pull are scaled by 1e-3 there).
"""
from __future__ import annotations

import numpy as np

F = np.float32


def heatmaps_loss(pred, target, mask):
    """loss.py:12-16: mean over B*K*h*w of (pred-target)^2 * mask[:,None].  -> (loss, dloss/dpred)"""
    d = (pred - target).astype(F)
    m = np.broadcast_to(mask[:, None], pred.shape)
    n = F(pred.size)
    return F((d * d * m).astype(np.float64).sum() / pred.size), (F(2) * d * m / n).astype(F)


def ae_grouping_loss(tags, joints):
    """loss.py:20-61 on tags [B,K,h,w] and joints = list over images of int [P,K,3] (x, y, vis).
    -> (push, pull, dpush/dtags, dpull/dtags), push/pull already divided by the batch size (before the 1e-3)."""
    B = len(joints)
    g_push, g_pull = np.zeros(tags.shape, np.float64), np.zeros(tags.shape, np.float64)
    push_loss = pull_loss = 0.0
    for i in range(B):
        refs, members, pull = [], [], 0.0
        for person in joints[i]:
            idx = [(k, int(y), int(x)) for k, (x, y, vis) in enumerate(person) if vis > 0]  # :26-30
            if not idx:
                continue
            t = np.array([tags[i, k, y, x] for k, y, x in idx], np.float64)
            m = t.mean()
            refs.append(m)
            members.append((idx, t, m))
            pull += ((t - m) ** 2).mean()  # :38
        n = len(refs)
        if n == 0:
            continue
        pull_loss += pull / n  # :45-48
        for idx, t, m in members:  # d mean((t-m)^2)/dt_k = 2 (t_k - m) / len  (the mean's own term sums to zero)
            for (k, y, x), tk in zip(idx, t):
                g_pull[i, k, y, x] += 2.0 * (tk - m) / len(idx) / n / B
        if n == 1:
            continue
        r = np.array(refs)
        d = r[:, None] - r[None, :]
        e = np.exp(-d * d)
        c = 0.5 / ((n - 1) * n)
        push_loss += (e.sum() - n) * c  # :57-60
        dm = (2.0 * (-2.0 * d * e)).sum(1) * c  # both (a,b) and (b,a) carry m_a
        for (idx, t, m), g in zip(members, dm):
            for k, y, x in idx:
                g_push[i, k, y, x] += g / len(idx) / B
    return F(push_loss / B), F(pull_loss / B), g_push.astype(F), g_pull.astype(F)


def calculate_loss(stage_preds, tags, stage_targets, masks, joints):
    """loss.py:71-93 -> (heatmap losses [2], push*1e-3, pull*1e-3, grads of their sum wrt (pred0, pred1, tags))."""
    hl, gp = [], []
    for p, t, m in zip(stage_preds, stage_targets, masks):
        l, g = heatmaps_loss(p, t, m)
        hl.append(l)
        gp.append(g)
    push, pull, gpush, gpull = ae_grouping_loss(tags, joints[0])
    gt = (F(1e-3) * (gpush.astype(np.float64) + gpull.astype(np.float64))).astype(F)
    return hl, F(push * F(1e-3)), F(pull * F(1e-3)), gp, gt
