"""Pin the CPU oracle to the golden vectors generated from the reference (tests/golden/make_golden.py)."""
import os
import numpy as np
import pytest
from oracle import pcgc_oracle as orc


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


from conftest import same_cpu_kind_as_golden as _same_cpu_kind


def test_g1_entropy_tables_bit_exact(golden_dir):
    """The reference-arithmetic table (the one the codec uses) equals the reference's own output bit for bit."""
    g = _load(golden_dir, 'entropy_tables.npz')
    if not _same_cpu_kind(g):
        pytest.skip('host CPU capability differs from the golden host (see _same_cpu_kind)')
    assert int(g['n_cases']) >= 15
    for ci in range(int(g['n_cases'])):
        params = g[f'c{ci}_params']
        lo, hi = g[f'c{ci}_minmax']
        cdf = orc.cdf_float_ref32(params, lo, hi)
        np.testing.assert_array_equal(cdf, g[f'c{ci}_cdf'], err_msg=f'case {ci}')
        np.testing.assert_array_equal(orc.cdf_table_ref32(params, lo, hi), orc.cdf_u16(g[f'c{ci}_cdf']))


def test_g1_product_host_table_bit_exact(golden_dir):
    """pcgcv2_amd.EntropyBottleneck.reference_table (host part of the product, no GPU involved) against the same golden:
    0 mismatching uint16 entries over all cases, incl. alphabets of 150-300 symbols."""
    import torch
    from pcgcv2_amd.entropy_model import EntropyBottleneck
    g = _load(golden_dir, 'entropy_tables.npz')
    if not _same_cpu_kind(g):
        pytest.skip('host CPU capability differs from the golden host (see _same_cpu_kind)')
    eb = EntropyBottleneck(8)
    total = 0
    for ci in range(int(g['n_cases'])):
        M, B, Fa = orc._eb_unpack(g[f'c{ci}_params'])
        with torch.no_grad():
            for dst, src in zip(list(eb._matrices) + list(eb._biases) + list(eb._factors), M + B + Fa):
                dst.copy_(src)
        lo, hi = g[f'c{ci}_minmax']
        cdf, q = eb.reference_table(lo, hi)
        np.testing.assert_array_equal(cdf.numpy(), g[f'c{ci}_cdf'], err_msg=f'case {ci}')
        np.testing.assert_array_equal(q, orc.cdf_u16(g[f'c{ci}_cdf']), err_msg=f'case {ci}')
        qn, cdfn = eb.reference_table_native(lo, hi, want_cdf=True)          # the C++ issue of the same operator sequence
        np.testing.assert_array_equal(cdfn, g[f'c{ci}_cdf'], err_msg=f'native, case {ci}')
        np.testing.assert_array_equal(qn, q, err_msg=f'native, case {ci}')
        total += q.size
    assert total > 8 * 1000


def test_native_and_python_tables_agree_on_any_host(golden_dir):
    """Whatever the host kind, the C++ and the Python issue of the operator sequence give the same table (same ATen kernels)."""
    import torch
    from pcgcv2_amd.entropy_model import EntropyBottleneck
    g = _load(golden_dir, 'entropy_tables.npz')
    eb = EntropyBottleneck(8)
    for ci in range(int(g['n_cases'])):
        M, B, Fa = orc._eb_unpack(g[f'c{ci}_params'])
        with torch.no_grad():
            for dst, src in zip(list(eb._matrices) + list(eb._biases) + list(eb._factors), M + B + Fa):
                dst.copy_(src)
        lo, hi = g[f'c{ci}_minmax']
        np.testing.assert_array_equal(eb.reference_table_native(lo, hi), eb.reference_table(lo, hi)[1])
        np.testing.assert_array_equal(eb.reference_table_native(lo, hi), orc.cdf_table_ref32(g[f'c{ci}_params'], lo, hi))


def test_g1_entropy_tables_fp64_evaluation(golden_dir):
    """The fp64 C evaluation (what the optional device kernel computes) stays within fp32 round-off of the reference."""
    g = _load(golden_dir, 'entropy_tables.npz')
    for ci in range(int(g['n_cases'])):
        params = g[f'c{ci}_params']
        lo, hi = g[f'c{ci}_minmax']
        lik = orc.likelihood(params, lo, hi)
        # reference evaluates in fp32 (torch CPU); the oracle in fp64 -> fp32: agree to fp32 round-off
        np.testing.assert_allclose(lik, g[f'c{ci}_likelihood'], rtol=2e-5, atol=3e-8)
        cdf = orc.cdf_float(params, lo, hi)
        np.testing.assert_allclose(cdf, g[f'c{ci}_cdf'], rtol=0, atol=2e-6)
        assert cdf.shape == g[f'c{ci}_cdf'].shape
        # the 16-bit tables built from either float cdf differ by at most 1 count
        t_ref = orc.cdf_u16(g[f'c{ci}_cdf']).astype(np.int64)
        t_orc = orc.cdf_u16(cdf).astype(np.int64)
        d = (t_ref - t_orc) % 65536
        assert np.all((d <= 1) | (d >= 65535))


def test_g2_ordering_and_topk(golden_dir):
    g = _load(golden_dir, 'ordering.npz')
    for i in range(3):
        c = g[f's{i}_coords']
        np.testing.assert_array_equal(orc.array2vector(c, c.max() + 1), g[f's{i}_key'])
        np.testing.assert_array_equal(orc.sort_zyx_perm(c), g[f's{i}_argsort'])
    for i in range(4):
        np.testing.assert_array_equal(orc.topk_mask(g[f't{i}_vals'], int(g[f't{i}_k'])), g[f't{i}_mask'])


def test_g3_ply_text(golden_dir, tmp_path):
    g = _load(golden_dir, 'ply_format.npz')
    assert orc.ply_ascii_bytes(g['coords']) == g['file_bytes'].tobytes()
    p = tmp_path / 'a.ply'
    p.write_bytes(g['file_bytes'].tobytes())
    np.testing.assert_array_equal(orc.read_ply_ascii_geo(str(p)), g['read_back'])
    p2 = tmp_path / 'b.ply'
    p2.write_bytes(g['file2_bytes'].tobytes())
    np.testing.assert_array_equal(orc.read_ply_ascii_geo(str(p2)), g['read_back2'])


def test_g4_d1_metric(golden_dir):
    g = _load(golden_dir, 'd1_metric.npz')
    for i in range(int(g['n_cases'])):
        m = orc.d1_metrics(g[f'p{i}_a'], g[f'p{i}_b'], int(g[f'p{i}_res']))
        # pc_error_d prints 6 significant digits
        assert m['mse1'] == pytest.approx(float(g[f'p{i}_mse1(p2point)']), rel=1e-5, abs=1e-9)
        assert m['mse2'] == pytest.approx(float(g[f'p{i}_mse2(p2point)']), rel=1e-5, abs=1e-9)
        assert m['mseF'] == pytest.approx(float(g[f'p{i}_mseF(p2point)']), rel=1e-5, abs=1e-9)
        if m['mseF'] > 0:
            assert m['psnrF'] == pytest.approx(float(g[f'p{i}_mseF_PSNR(p2point)']), abs=2e-4)


def test_range_coder_roundtrip_and_known_answer():
    rng = np.random.default_rng(0)
    params = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'entropy_tables.npz'))['c1_params']
    for n, lo, hi in [(1, -1, 1), (7, 0, 0), (1000, -6, 7), (20000, -20, 20)]:
        table = orc.cdf_table_ref32(params, lo, hi)
        L = hi - lo + 1
        assert table.shape == (8, L + 1)
        # strictly increasing except the wrapped last entry (torchac forces c_high = 0x10000 for the max symbol)
        assert np.all(np.diff(table[:, :-1].astype(np.int64), axis=1) > 0)
        sym = rng.integers(0, L, size=(n, 8)).astype(np.int16)
        data = orc.rc_encode(table, sym)
        back = orc.rc_decode(table, data, sym.size).reshape(n, 8)
        np.testing.assert_array_equal(back, sym)
    # known answer, derivable by hand from the published algorithm: a single symbol 0 of a 2-symbol alphabet with
    # c_low=0,c_high=0x8000 -> high<0x80000000 emits '0', then the terminating pending+1 with low<0x40000000 emits '0','1'
    t = np.array([[0, 0x8000, 0]], np.uint16)
    assert orc.rc_encode(t, np.array([0], np.int16)) == bytes([0b00100000])
