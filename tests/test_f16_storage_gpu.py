"""The f16 storage build of the kernel library (libmammoclip_hip_f16.so, MC_STORAGE=f16): the reference's AMP dtype
[ref: trainer.py:271-278 -- fp16 autocast + GradScaler], and the configuration in which north_star's |loss - reference|
<= 1e-3 holds in TRAIN mode (VERDICT r3 "what is missing" #2).  The storage type is fixed when the library is loaded, so
every test here drives a child process."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a HIP device", allow_module_level=True)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _run(args, timeout, **extra_env):
    env = dict(os.environ, MC_STORAGE="f16", **extra_env)
    p = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    return p


def _worker(which, timeout=600):
    p = _run([os.path.join(HERE, "_f16_worker.py"), which], timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("F16-WORKER ")][-1]
    return json.loads(line[len("F16-WORKER "):])


def test_f16_build_every_kernel_test():
    """tests/test_kernels_gpu.py -- every entry point of the C ABI against its torch fp32 reference -- on the f16 build
    (the tests draw their 16-bit inputs in the loaded library's storage type)"""
    p = _run(["-m", "pytest", os.path.join(HERE, "test_kernels_gpu.py"), "-m", "gpu", "-x", "-q"], 1200)
    assert p.returncode == 0, p.stdout[-3000:]
    assert " passed" in p.stdout and "failed" not in p.stdout, p.stdout[-1000:]


def test_f16_model_kats_and_config1_vs_reference():
    """the reference-generated fixtures of the model level on the f16 build: MBConv / BERT / loss known-answer tests, the
    config-#1 end-to-end fixtures (eval |loss - reference| <= 1e-3 on three seeds; train mode, gradients)"""
    p = _run(["-m", "pytest", os.path.join(HERE, "test_model_gpu.py"), "-m", "gpu", "-x", "-q", "-k",
              "mbconv_kats or bert_kat or loss_kats or config1_eval or e2e_vs_reference"], 1200)
    assert p.returncode == 0, p.stdout[-3000:]
    assert " passed" in p.stdout and "failed" not in p.stdout, p.stdout[-1000:]


def test_f16_bn8k_fixture_train_and_eval_within_1e3():
    """The reference's own bn8k fixture (B2 + BERT-base, b = 8, 456^2, T = 64: >= 1800 samples per BatchNorm channel).  bf16
    storage: eval -1.3e-3, TRAIN +5.4e-3, gradient cosines >= 0.980, norms within 4 % (test_train_mode_vs_reference_bn8k_fixture).
    f16 storage, measured on MI355X: eval +7e-5, train +5e-5, embedding cosines >= 0.999994, gradient cosines >= 0.9996, norms
    within 0.6 %.  Bounds: north_star's 1e-3 on both losses (and 3e-4 as the regression bound of this build), cosines 0.9999 /
    0.999, norms 1.5 %."""
    r = _worker("bn8k")
    print("f16 bn8k", r)
    assert abs(r["eval_dloss"]) <= 1e-3 and abs(r["train_dloss"]) <= 1e-3, r
    assert abs(r["eval_dloss"]) <= 3e-4 and abs(r["train_dloss"]) <= 3e-4, r
    assert r["eval_min_cos"] >= 0.9999 and r["train_min_cos"] >= 0.9999, r
    assert r["n_grads"] >= 10 and r["nonfinite_grads"] == 0, r
    assert r["grad_min_cos"] >= 0.999 and 0.985 <= r["grad_norm_ratio_min"] and r["grad_norm_ratio_max"] <= 1.015, r


def test_f16_trainer_dynamic_loss_scale():
    """engine.LossScaler = GradScaler's policy [ref: trainer_ddp.py:296-303]: auto-installed on the f16 build at 65536; a clean
    step steps (unscaled, finite gradients that agree with an unscaled backward), an overflowing step is skipped -- no
    parameter, no optimizer step counter moves -- and halves the scale"""
    r = _worker("scaler")
    print("f16 scaler", r)
    assert r["auto_scaler"] == "LossScaler" and r["init_scale"] == 65536.0, r
    assert r["clean_skipped"] == 0 and r["clean_changed"] > 400 and r["clean_grads_finite"], r
    assert r["scaled_vs_unscaled_cos"] >= 0.999, r
    assert r["overflow_skipped"] == 1 and r["overflow_scale_after"] == 2.0 ** 39, r
    assert r["overflow_params_unchanged"] and r["overflow_steps_unchanged"], r
    assert r["overflow_reported_skipped"] == 1 and r["overflow_reported_scale"] == 2.0 ** 39, r
    # skip-then-recover under the DYNAMIC policy: from 2^24 the scale halves on every skipped step until the
    # gradients fit f16, then stays; every later step applies -- skipped + applied = 14, at least one of each, and the
    # scale sequence is non-increasing by exact factors of two
    seq = r["recover_seq"]
    n_skip = seq[-1][1]
    assert 1 <= n_skip <= 12 and r["recover_applied"] == 14 - n_skip and r["recover_params_finite"], r
    # every skip halves the scale exactly once, nothing else moves it (no growth inside 2000 clean steps); the skips need not be
    # consecutive -- the parameters move between steps (measured: skips at steps 1-3 and again at step 6, 2^24 -> 2^20)
    assert all(sc == 2.0 ** (24 - sk) for sc, sk in seq) and all(a[1] <= b[1] for a, b in zip(seq, seq[1:])), r
    assert seq[0][1] == 1, r                                   # 2^24 overflows f16 on the first step
    assert r["scaler_state_roundtrip"], r
    assert r["static_dynamic_flag"] is False and r["static_scale_after"] == 512.0 and r["static_skipped"] == 0, r


def test_f16_baseline_shapes_train_and_eval_within_1e3_of_the_oracle():
    """north_star's bound at the BASELINE per-sample shapes, BOTH modes.  Measured on MI355X (f16 | bf16 storage):
    cfg3 / cfg4 shape (B5, 1520 x 912, T = 256) eval 3e-5 | 4.4e-4, train 2.0e-4 | 3.2e-3; cfg2 shape (B2, 912^2) eval
    5e-5 | 1.8e-4, train 2.5e-4 | 3.4e-4."""
    r = _worker("shapes", timeout=900)
    print("f16 shapes", r)
    for tag in ("cfg3", "cfg2"):
        assert abs(r[tag + "/eval_dloss"]) <= 1e-3 and abs(r[tag + "/train_dloss"]) <= 1e-3, r
        assert r[tag + "/eval_min_cos"] >= 0.99999 and r[tag + "/train_min_cos"] >= 0.9999, r
    # round 5: the BACKWARD at 1520 x 912 under the dynamic loss scale at its default 65536 -- no overflow (the step would
    # apply), and the sampled parameter gradients agree with the fp32 oracle far better than the bf16 build's (image 0.87-0.95,
    # norms within 4 % there).  Measured on MI355X (profiles/r05_f16_storage_parity.txt): cosines 0.99848 (_blocks.21._se_reduce)
    # .. 0.99998 (text), norm ratios 0.9921 .. 1.0025, logit_scale 1.0136 -- floors 0.997 / 1.5 % (logit_scale 3 %)
    assert r["cfg3/bwd_finite"] and r["cfg3/bwd_skipped"] == 0 and r["cfg3/bwd_scale_after"] == 65536.0, r
    assert r["cfg3/grad_min_cos"] >= 0.997, r
    assert all(abs(v - 1.0) <= (0.03 if k == "logit_scale" else 0.015) for k, v in r["cfg3/grad_norm_ratio"].items()), r


def test_f16_trainer_level_tests_under_the_dynamic_loss_scale():
    """the Trainer-level tests on the f16 build under the DYNAMIC scaler (round 5; round 4 pinned a static scale here): every
    ``Trainer(loss_scale="auto")`` gets ``LossScaler(init_scale=256)`` through MC_LOSS_SCALE_INIT -- the device-side policy
    (unscale by the device scalar, flag read by the AdamW kernel, update kernel) runs in every step; 256 is low enough that no
    step of these small configurations skips, so single steps and the 4-step trajectory stay comparable with the reference's
    (a scaler that starts at 65536 skips its first steps there, as GradScaler does: test_f16_trainer_dynamic_loss_scale asserts
    that skip-then-recover sequence).  Covered: the two-rank steps over gloo on a shared GPU (gradients are unscaled AFTER
    the rank average, bit-identical on both ranks), the 4-step trajectory against the reference's own loop, the evaluator /
    checkpoint entry points."""
    p = _run(["-m", "pytest", os.path.join(HERE, "test_dist_gpu.py"), os.path.join(HERE, "test_model_gpu.py"), "-m", "gpu", "-x", "-q",
              "-k", "two_rank_step_equals or trajectory or evaluator"], 1500, MC_LOSS_SCALE_INIT="256")
    assert p.returncode == 0, p.stdout[-3000:]
    assert " passed" in p.stdout and "failed" not in p.stdout, p.stdout[-1000:]
