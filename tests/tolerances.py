"""Parity tolerances of the -m gpu tests that compare a whole step with the CPU oracle (the same numbers as bench.py's same-run parity
record).  About 10x what is measured on the MI355X: losses 1.9e-6 .. 3.8e-6 on |16.1|, logits 1.4e-6 .. 1.7e-6 on |1.2|, worst
parameter-gradient tensor 1.4e-5 (batch 64) .. 6e-5 (batch 2) of its maximum (profiles/r04_bench.json, profiles/r04_pytest_gpu.log).

  losses            |d| <= 2e-6 + 2e-6 * |ref|      (fp32 on both sides, different summation orders)
  logits            |d| <= 1e-5 + 1e-5 * max|ref|
  gradient tensors  max|d| <= 2e-4 * max|ref|       (train.py:265-266: what optimizer.minimize differentiates)
  tokens            exact; the oracle's minimum top-2 logit margin must exceed MARGIN_FACTOR x the logit tolerance where a test
                    chooses its seeds (two logits may move towards each other by one tolerance each, factor 2 to spare)
"""
LOSS_ATOL = LOSS_RTOL = 2e-6
LOGIT_ATOL = LOGIT_RTOL = 1e-5
GRAD_RTOL = 2e-4
MARGIN_FACTOR = 4.0


def loss_tol(ref):
    return LOSS_ATOL + LOSS_RTOL * abs(float(ref))


def logit_tol(max_abs_ref):
    return LOGIT_ATOL + LOGIT_RTOL * abs(float(max_abs_ref))
