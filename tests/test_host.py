"""The C++ host mirror (ftrl-ffm_amd/host): builds with g++ against libffm_engine.so; its own test
executable restates the reference's doctest cases; the CLI reproduces the reference's printed
losses on the bundled data."""
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

import ftrl_ffm_amd as fa
from util import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_builds_and_cpu_side_tests_pass(tmp_path):
    _, test_bin = fa.build_host()
    out = subprocess.run([test_bin, "cpu"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failed" in out.stdout


def test_cli_rejects_unknown_flag(tmp_path):
    main_bin, _ = fa.build_host()
    out = subprocess.run([main_bin, "--epoch", "3"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode != 0 and "invalid argument" in out.stderr


@pytest.mark.gpu
def test_host_gpu_side_tests_pass(tmp_path):
    _, test_bin = fa.build_host()
    out = subprocess.run([test_bin, "gpu"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failed" in out.stdout


def _bundled(tmp_path, libsvm=False):
    with gzip.open(os.path.join(GOLDEN, "data", "libffm_data.txt.gz"), "rt") as f:
        text = f.read()
    if libsvm:
        text = "\n".join(" ".join([t.split()[0]] + [tok.split(":", 1)[1] for tok in t.split()[1:]])
                         for t in text.splitlines()) + "\n"
    p = tmp_path / ("libsvm_data.txt" if libsvm else "libffm_data.txt")
    p.write_text(text)
    return str(p)


@pytest.mark.gpu
@pytest.mark.parametrize("online", ["true", "false"])
def test_cli_ffm_on_bundled_data_prints_the_reference_losses(tmp_path, online):
    """Reference CLI, FFM defaults, 3 epochs on data/libffm_data.txt prints train/eval loss
    0.6907/0.6893, 0.6883/0.6874, 0.6867/0.6860.  Online mode keeps file
    order, so with the block ramp every epoch must be within 1e-4; offline shuffles, so only the
    first digits are comparable (2e-3)."""
    main_bin, _ = fa.build_host()
    data = _bundled(tmp_path)
    out = subprocess.run([main_bin, "--train_data", data, "--eval_data", data, "--model_type", "FFM",
                          "--n_epochs", "3", "--online", online, "--batch_size", "256"],
                         cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    tr = [float(x) for x in re.findall(r"train loss: ([0-9.]+)", out.stdout)]
    ev = [float(x) for x in re.findall(r"eval loss: ([0-9.]+)", out.stdout)]
    z = np.load(os.path.join(GOLDEN, "g9_bundled_ffm_end_to_end.npz"))
    assert len(tr) == 3 and len(ev) == 3
    tol = 1e-4 + 5e-5 if online == "true" else 2e-3   # +5e-5: the CLI prints 4 decimals
    assert np.allclose(tr, z["train_loss"], atol=tol), (tr, z["train_loss"])
    assert np.allclose(ev, z["eval_loss"], atol=tol), (ev, z["eval_loss"])


@pytest.mark.gpu
def test_cli_config1_lr_on_libsvm(tmp_path):
    """BASELINE.json configs[0]: LR FTRL on data/libsvm_data.txt, one epoch: 0.6907 / 0.6893."""
    main_bin, _ = fa.build_host()
    data = _bundled(tmp_path, libsvm=True)
    out = subprocess.run([main_bin, "--train_data", data, "--eval_data", data, "--model_type", "LR",
                          "--n_epochs", "1", "--online", "true", "--batch_size", "256"],
                         cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    tr = float(re.findall(r"train loss: ([0-9.]+)", out.stdout)[0])
    ev = float(re.findall(r"eval loss: ([0-9.]+)", out.stdout)[0])
    assert abs(tr - 0.6907) < 2e-4 and abs(ev - 0.6893) < 2e-4


@pytest.mark.gpu
def test_cli_n_gpus_shards_match_one_engine(tmp_path):
    """--n_gpus N (one field-pair shard engine per device, one all-reduce of the partial logits per
    block: ffm_group_*).  On a one-GPU box the shards share the device (FTRL_SAME_DEVICE=1: the
    collective is then a device-local sum, the orchestration is the N-GPU one): the printed epoch
    losses must be the one-engine run's to the printed precision; and with FFM_GROUP_RCCL=1 a group
    of ONE runs the librccl ncclAllReduce on the engine's stream -- same losses again."""
    from ftrl_ffm_amd import synth
    main_bin, _ = fa.build_host()
    F, per = 12, 500
    g = synth.Generator(F, F * per, "zipf", seed=5)
    data = tmp_path / "s.ffm"
    data.write_text(synth.to_libffm_text(g.block(20000)))
    base = [main_bin, "--train_data", str(data), "--eval_data", str(data), "--model_type", "FFM", "--n_fields", str(F),
            "--n_feats", str(F * per), "--n_factors", "8", "--online", "true", "--n_epochs", "2", "--batch_size", "1024",
            "--w_alpha", "0.05", "--w_l1", "0.01", "--w_l2", "0.1", "--field_ranges", "uniform"]
    def losses(extra, env):
        out = subprocess.run(base + extra, cwd=tmp_path, capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, **env))
        assert out.returncode == 0, out.stdout + out.stderr
        return ([float(x) for x in re.findall(r"train loss: ([0-9.]+)", out.stdout)],
                [float(x) for x in re.findall(r"eval loss: ([0-9.]+)", out.stdout)], out.stdout)
    tr1, ev1, _ = losses([], {})
    assert len(tr1) == 2 and len(ev1) == 2 and tr1[1] < tr1[0] < 0.6931
    for n in (2, 4):
        tr, ev, text = losses(["--n_gpus", str(n)], {"FTRL_SAME_DEVICE": "1"})
        assert "%d field-pair shards" % n in text
        assert np.allclose(tr, tr1, atol=2e-4) and np.allclose(ev, ev1, atol=2e-4), (n, tr, tr1, ev, ev1)
    tr, ev, text = losses(["--n_gpus", "1"], {"FFM_GROUP_RCCL": "1"})
    assert tr == tr1 and ev == ev1


from oracle import pyoracle  # noqa: E402

needs_ref = pytest.mark.skipif(not pyoracle.have_ref(), reason="oracle/_ref not built here")


@needs_ref
def test_model_files_interoperate_with_the_reference(tmp_path):
    """Files written by the reference (ffm.cpp:138-180) are read by the host's file functions and
    vice versa: zstd frames exactly, the text format to its printed precision."""
    from oracle.pyoracle import CpuModel
    from util import rand_state
    pyoracle.build(ref=True)
    _, test_bin = fa.build_host()
    nf, F, k = 30, 3, 4
    rng = np.random.default_rng(3)
    ref = CpuModel("ref", "FFM", nf, F, k)
    st = rand_state(rng, ref)
    ref.set_state(st)
    zst, txt = str(tmp_path / "ref.zst"), str(tmp_path / "ref.txt")
    assert ref.lib.fr_save_model(ref.h, zst.encode(), 1, 5) == 0
    assert ref.lib.fr_save_model(ref.h, txt.encode(), 0, 0) == 0
    # reference zst -> host -> text and zst again
    mine_txt, mine_zst = str(tmp_path / "mine.txt"), str(tmp_path / "mine.zst")
    for src, dst in ((zst, mine_zst), (zst, mine_txt), (txt, str(tmp_path / "from_txt.zst"))):
        out = subprocess.run([test_bin, "convert", src, dst, str(nf), str(F * k)], cwd=tmp_path,
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
    # host-written zst read back by the reference: exact weights
    ref2 = CpuModel("ref", "FFM", nf, F, k)
    assert ref2.lib.fr_load_model(ref2.h, mine_zst.encode(), 1) == 0
    got = ref2.get_state()
    for key in ("bias3", "lin_w", "vec_w"):
        want = st[key].copy()
        if key == "bias3":
            assert got[key][0] == want[0]
        else:
            assert np.array_equal(got[key], want), key
    # host-written text read back by the reference: vec_w exactly (9 significant digits), bias and
    # lin_w to the 6 digits an ostream prints
    ref3 = CpuModel("ref", "FFM", nf, F, k)
    assert ref3.lib.fr_load_model(ref3.h, mine_txt.encode(), 0) == 0
    got = ref3.get_state()
    assert np.array_equal(got["vec_w"], st["vec_w"])
    assert np.allclose(got["lin_w"], st["lin_w"], rtol=1e-5)
    # reference-written text -> host zst -> reference: what the text held, exactly
    ref4, ref5 = CpuModel("ref", "FFM", nf, F, k), CpuModel("ref", "FFM", nf, F, k)
    assert ref4.lib.fr_load_model(ref4.h, str(tmp_path / "from_txt.zst").encode(), 1) == 0
    assert ref5.lib.fr_load_model(ref5.h, txt.encode(), 0) == 0
    a, b = ref4.get_state(), ref5.get_state()
    assert np.array_equal(a["vec_w"], b["vec_w"]) and np.array_equal(a["lin_w"], b["lin_w"])
