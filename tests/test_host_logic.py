"""CPU: host-side logic around the kernels (sharding, synthetic data, label canonicalisation, parameter
formats, state-dict compatibility)."""
import numpy as np
import pytest
import torch


def test_shard_range_partitions_exactly():
    from sednet_hip.shard import shard_range
    for n in (0, 1, 7, 64, 512, 513):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_synthetic_cloud_contract():
    from sednet_hip import synth
    p, n, l, t = synth.synthetic_cloud(3, 2000)
    assert p.shape == (2000, 3) and p.dtype == np.float32 and n.shape == (2000, 3)
    np.testing.assert_allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-5)
    np.testing.assert_allclose(p.mean(0), 0, atol=1e-5)
    assert 0.5 < np.max(p.max(0) - p.min(0)) < 1.8        # unit-scaled before the PCA rotation (dataset_segments.py:400-417)
    assert np.var(p[:, 0]) <= np.var(p[:, 1]) + 1e-6 <= np.var(p[:, 2]) + 2e-6   # PCA: smallest axis -> x
    assert set(np.unique(t)) <= {1, 3, 4, 5} and np.bincount(l).min() >= 24
    p2, *_ = synth.synthetic_cloud(3, 2000)
    np.testing.assert_array_equal(p, p2)


def test_closed_form_weights_match_reference_key_set():
    from sednet_hip import synth
    from src.SEDNet import SEDNet
    sd = synth.closed_form_state_dict(0)
    m = SEDNet(embedding=True, emb_size=128, primitives=True, num_primitives=6, mode=5, num_channels=6,
               combine_label_prim=True, edge_module=True, late_fusion=True, nn_nb=20)
    # SURVEY.md section 5 key set (+ the persistent buffer of the unused positional encoding, present in real checkpoints)
    assert sorted(m.state_dict().keys()) == sorted(list(sd.keys()) + ["pos_enc.inv_freq"])
    assert tuple(m.state_dict()["pos_enc.inv_freq"].shape) == (128,)
    assert sum(p.numel() for p in m.parameters()) == 1351432
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    # "module."-prefixed DataParallel checkpoints load after the reference's prefix strip (generate_predictions_aug.py:192)
    pref = {"module." + k: torch.from_numpy(v) for k, v in sd.items()}
    stripped = {k[k.find(".") + 1:]: v for k, v in pref.items()}
    m.load_state_dict(stripped, strict=True)
    with_buf = dict(stripped, **{"pos_enc.inv_freq": torch.zeros(128), "pos_enc.cached_penc": torch.zeros(1)})
    m.load_state_dict(with_buf, strict=True)            # checkpoints from the real positional_encodings package
    np.testing.assert_array_equal(sd["encoder.bn1.weight"], sd["encoder.conv1.1.weight"])   # aliased GroupNorm
    assert (sd["encoder.bn1.weight"] < 0).any()                        # exercises the min-over-k branch


def test_unsupported_configurations_fail_loudly():
    from src.SEDNet import SEDNet
    with pytest.raises(NotImplementedError):
        SEDNet(embedding=True, primitives=True, predict_normal=True)
    m = SEDNet(embedding=True, emb_size=128, primitives=True, num_primitives=6, mode=5, num_channels=6, nn_nb=20)
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 6, 64))


def test_canonical_labels_and_seg_iou():
    from oracle.mean_shift import canonical_labels
    from src.segment_utils import seg_iou
    a = np.array([5, 5, 2, 2, 9, 5])
    np.testing.assert_array_equal(canonical_labels(a), [0, 0, 1, 1, 2, 0])
    b = np.array([1, 1, 0, 0, 7, 1])
    np.testing.assert_array_equal(canonical_labels(a), canonical_labels(b))
    assert seg_iou(a, b) == 1.0
    c = b.copy(); c[0] = 0
    assert 0.5 < seg_iou(a, c) < 1.0


def test_parameter_entry_format():
    from src.primitive_forward import params_to_entry
    q = torch.arange(8, dtype=torch.float32)
    e = params_to_entry(1, q)
    assert e[0] == "plane" and tuple(e[1].shape) == (3, 1) and e[2].dim() == 0
    e = params_to_entry(3, q)
    assert e[0] == "cone" and tuple(e[1].shape) == (1, 3) and tuple(e[2].shape) == (3, 1)
    e = params_to_entry(4, q)
    assert e[0] == "cylinder" and tuple(e[1].shape) == (3, 1) and tuple(e[2].shape) == (1, 3)
    e = params_to_entry(5, q)
    assert e[0] == "sphere" and tuple(e[1].shape) == (1, 3)


def test_to_one_hot_and_weights_normalize_cpu(golden):
    from src.fitting_utils import to_one_hot, weights_normalize
    g = golden("f_fit")
    np.testing.assert_array_equal(to_one_hot(torch.from_numpy(g["oh_in"].astype(np.int64)), 7).numpy(), g["oh_out"])
    np.testing.assert_allclose(weights_normalize(torch.from_numpy(g["wn_in"]), 0.3).numpy(), g["wn_out"], rtol=1e-5, atol=1e-7)


def test_guard_loop_quantile_schedule():
    """int(quantile * num_samples) in Python double arithmetic (mean_shift.py:132): the third pass is 215, not
    216, because 0.015 * 1.2 * 1.2 * 10000 = 215.99999999999997 -- the K the mirror must (and does) use."""
    q, ks = 0.015, []
    for _ in range(4):
        ks.append(int(q * 10000)); q *= 1.2
    assert ks == [150, 180, 215, 259]


def test_src_package_shadows_and_falls_back(tmp_path):
    """with sed-net_amd ahead on sys.path the mirrors answer `from src.X import ...`; appending the
    reference's src directory to src.__path__ lets every module the mirrors do not provide keep resolving there."""
    import importlib
    import src
    ref_src = tmp_path / "src"
    ref_src.mkdir()
    (ref_src / "dataset_segments.py").write_text("ori_simple_data = 'reference loader'\n")
    (ref_src / "mean_shift.py").write_text("MeanShift = 'reference implementation (must stay shadowed)'\n")
    src.__path__.append(str(ref_src))
    try:
        ds = importlib.import_module("src.dataset_segments")
        assert ds.ori_simple_data == "reference loader"
        ms = importlib.import_module("src.mean_shift")
        assert "sed-net_amd" in ms.__file__ and not isinstance(ms.MeanShift, str)
    finally:
        src.__path__.remove(str(ref_src))
        import sys
        sys.modules.pop("src.dataset_segments", None)
