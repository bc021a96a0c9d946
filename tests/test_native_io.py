"""Native results files (post/NativeIO.hpp): the C-ABI writer / reader (host code, no GPU) against the numpy restatement
of the format in oracle/oracle_np.py, in both directions; slices written by several ranks in any order; the reference's
save -> load round trip of tests/SaveLoadTests.cpp (fields set from analytic functions, loaded back through the node-id
list of a different partition, compared to the functions); error behaviour."""
import os
import sys
import threading

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import oracle_np as ONP  # noqa: E402
from l3ster_amd import capi, native_io, system  # noqa: E402


def test_writer_matches_format_restatement(tmp_path):
    rng = np.random.default_rng(3)
    fields = rng.standard_normal((3, 1001))
    path = tmp_path / "a.res"
    native_io.save(path, fields, 1001, comment="two\nlines")
    assert path.read_bytes() == ONP.results_file_bytes(fields, "two\nlines")  # bit-exact, newline in the comment replaced
    comment, vals = ONP.results_file_parse(path.read_bytes())
    assert comment == "two lines" and np.array_equal(vals, fields)


def test_reader_reads_restated_file(tmp_path):
    rng = np.random.default_rng(4)
    fields = rng.standard_normal((2, 500))
    path = tmp_path / "b.res"
    path.write_bytes(ONP.results_file_bytes(fields, "made by the restatement"))
    assert native_io.info(path) == (2, 500)
    assert np.array_equal(native_io.load(path, 1), fields[1])
    assert np.array_equal(native_io.load(path, 0, node_begin=100, n=50), fields[0, 100:150])
    ids = rng.permutation(500)[:200]  # Loader::loadResultsImpl: dest(i) = results(old_node[i], field)
    assert np.array_equal(native_io.load(path, 1, node_ids=ids), fields[1, ids])
    ids = np.r_[np.arange(10, 60), [3], np.arange(400, 500)]  # runs of consecutive ids
    assert np.array_equal(native_io.load(path, 0, node_ids=ids), fields[0, ids])


@pytest.mark.parametrize("order_of_ranks", [(0, 1, 2, 3), (3, 1, 0, 2)])
def test_slices_of_several_ranks_any_order(tmp_path, order_of_ranks):
    fields = np.random.default_rng(5).standard_normal((2, 1000))
    bounds = [0, 137, 500, 501, 1000]
    path = tmp_path / "c.res"
    for r in order_of_ranks:
        native_io.save(path, fields[:, bounds[r]:bounds[r + 1]], 1000, bounds[r], comment="parallel", write_header=(r == 0))
    assert path.read_bytes() == ONP.results_file_bytes(fields, "parallel")


def test_concurrent_rank_writers(tmp_path):
    fields = np.random.default_rng(6).standard_normal((4, 20000))
    bounds = np.linspace(0, 20000, 9).astype(int)
    path = tmp_path / "d.res"
    ts = [threading.Thread(target=native_io.save, args=(path, fields[:, bounds[r]:bounds[r + 1]], 20000, bounds[r], "x", r == 0))
          for r in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert path.read_bytes() == ONP.results_file_bytes(fields, "x")


def test_save_load_round_trip_across_partitions(tmp_path):
    """tests/SaveLoadTests.cpp: fields sin(x/2), cos(y/2) set at the nodes, saved from a 2x2x1 partition, loaded into a
    1x1x2 partition through its nodes' ids in the saved numbering, compared with the functions."""
    ne, p = (4, 4, 4), 2

    def f(xyz):
        return np.stack([np.sin(0.5 * xyz[:, 0]), np.cos(0.5 * xyz[:, 1])])

    path = tmp_path / "parallel.res"
    saved_id_of_grid = {}
    for r in range(4):
        part = system.CubePartition(ne, p, parts=(2, 2, 1), rank=r)
        xyz = part.node_coords()[:part.n_owned_nodes]
        gid = part.node_grid_id[:part.n_owned_nodes]
        for i, g in enumerate(gid):
            saved_id_of_grid[int(g)] = part.global_node_base + i
        # node-interleaved "solution vector" with 3 dofs per node; dofs 0 and 2 are saved
        x = np.zeros((part.n_owned_nodes, 3))
        x[:, 0], x[:, 2] = f(xyz)
        native_io.save_solution(path, x.reshape(-1), part, 3, dof_inds=(0, 2), comment="SaveLoadTests")
    assert native_io.info(path) == (2, 9 ** 3)
    for r in range(2):
        part = system.CubePartition(ne, p, parts=(1, 1, 2), rank=r)
        n_local = part.n_owned_nodes + part.n_ghost_nodes
        old_ids = np.array([saved_id_of_grid[int(g)] for g in part.node_grid_id[:n_local]])
        want = f(part.node_coords()[:n_local])
        for k in range(2):
            got = native_io.load(path, k, node_ids=old_ids)
            assert np.max(np.abs(got - want[k])) < 1e-6  # the reference's threshold; here exact
            assert np.array_equal(got, want[k])


def test_errors_are_reported(tmp_path):
    path = tmp_path / "e.res"
    native_io.save(path, np.ones((1, 10)), 10)
    with pytest.raises(capi.L3KError):
        native_io.load(path, 1)  # field index out of range (throwingAssert(max(src_inds) < results.fields()))
    with pytest.raises(capi.L3KError):
        native_io.load(path, 0, node_ids=[0, 10])  # node id outside the file
    with pytest.raises(capi.L3KError):
        native_io.load(path, 0, node_begin=5, n=6)
    with pytest.raises(capi.L3KError):
        native_io.info(tmp_path / "missing.res")
    (tmp_path / "bad.res").write_bytes(b"not a results file")
    with pytest.raises(capi.L3KError):
        native_io.info(tmp_path / "bad.res")
    (tmp_path / "short.res").write_bytes(ONP.results_file_bytes(np.ones((2, 10)))[:-8])
    with pytest.raises(capi.L3KError):
        native_io.info(tmp_path / "short.res")
    with pytest.raises(capi.L3KError):
        native_io.save(path, np.ones((1, 10)), 5)  # slice larger than the file's node count
    with pytest.raises(capi.L3KError):
        native_io.save(path, np.ones((0, 10)), 10)  # no fields (throwingAssert(not inds.empty()))
