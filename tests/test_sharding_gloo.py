"""CPU, world_size 2 (and 3) over gloo: the row sharder that splits each fused model batch across ranks and
all-gathers the outputs must hand every rank exactly the tensor a single process computes, for even and ragged
row counts, with per-row side inputs (text / pooled / condition rows)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from elasticdiffusion_official_amd.sharding import RowSharder, row_partition
from tests.procs import join_all, single_thread


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model(x, text, pooled, cond):
    """Row-wise deterministic stand-in for the UNet: every output row depends only on its own inputs."""
    y = torch.tanh(x * 1.5) + x.flip(-1) * 0.25
    if text is not None:
        y = y + text.mean(dim=(1, 2)).view(-1, 1, 1, 1)
    if pooled is not None:
        y = y * (1 + 0.1 * pooled.sum(dim=1).view(-1, 1, 1, 1))
    if cond is not None:
        y = y + cond.mean(dim=(1, 2, 3)).view(-1, 1, 1, 1)
    return y


def _worker(rank, world, port, n_rows_list, ret):
    single_thread()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = RowSharder()
        assert (sh.world_size, sh.rank) == (world, rank)
        ok = True
        for n in n_rows_list:
            g = torch.Generator().manual_seed(n)
            x = torch.randn(n, 4, 8, 8, generator=g)
            text = torch.randn(n, 5, 6, generator=g)
            pooled = torch.randn(n, 3, generator=g)
            cond = torch.randn(n, 3, 16, 16, generator=g)
            calls = []

            def fn(a, b, c, d):
                calls.append(a.shape[0])
                return _model(a, b, c, d)

            for with_side in (True, False):
                full = sh.run(fn, x, text if with_side else None, pooled if with_side else None, cond if with_side else None)
                want = _model(x, text if with_side else None, pooled if with_side else None, cond if with_side else None)
                ok = ok and torch.equal(full, want)
            _, spans = row_partition(n, world)
            own = spans[rank][1] - spans[rank][0]
            # a rank runs the model on exactly the rows it owns: no duplicated rows, no call at all when it owns none
            ok = ok and calls == ([own, own] if own else [])
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharder_matches_single_process(world):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, [20, 6, 1, 7, 24], ret)) for r in range(world)]
    for p in procs:
        p.start()
    join_all(procs, 120, "row sharder over gloo")
    assert all(ret.get(r) is True for r in range(world)), dict(ret)


def test_row_partition_covers_rows_once():
    for n in range(1, 40):
        for ws in (1, 2, 3, 4, 8):
            per, spans = row_partition(n, ws)
            assert per * ws >= n and len(spans) == ws
            covered = [i for a, b in spans for i in range(a, b)]
            assert covered == list(range(n))
            assert all(b - a <= per for a, b in spans)


def test_single_process_is_identity():
    sh = RowSharder()
    x = torch.randn(5, 4, 8, 8)
    assert torch.equal(sh.run(_model, x, None, None, None), _model(x, None, None, None))


def _group_worker(rank, world, port, g, ret):
    single_thread()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        groups = [dist.new_group(ranks=list(range(i * g, (i + 1) * g))) for i in range(world // g)]
        sh = RowSharder(groups[rank // g])
        ok = (sh.world_size, sh.rank) == (g, rank % g)
        x = torch.randn(7, 4, 8, 8, generator=torch.Generator().manual_seed(100 + rank // g))  # per-group data
        ok = ok and torch.equal(sh.run(_model, x, None, None, None), _model(x, None, None, None))
        solo = RowSharder(False)
        ok = ok and (solo.world_size, solo.rank) == (1, 0) and torch.equal(solo.run(_model, x), _model(x, None, None, None))
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def test_sub_groups_shard_independently():
    """bench.py's N-GPU layout: N/g groups of g ranks, each group row-shards its own image."""
    world, g = 4, 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_group_worker, args=(r, world, port, g, ret)) for r in range(world)]
    for p in procs:
        p.start()
    join_all(procs, 120, "row sharder over gloo")
    assert all(ret.get(r) is True for r in range(world)), dict(ret)


def _mismatch_worker(rank, world, port, ret):
    single_thread()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = RowSharder()
        x = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(5))  # one row on two ranks: rank 1 owns nothing and must guess the output row shape

        def fn(a, *_):
            return a.mean(dim=(2, 3))  # changes the row shape: [1,4,8,8] -> [1,4]

        try:
            sh.run(fn, x)               # no out_like: rank 1 falls back to the input row shape -> mismatch
            ret[rank] = "no error"
        except RuntimeError as e:
            ret[rank] = "raised" if "out_like" in str(e) else f"other: {e}"
        full = sh.run(fn, x, out_like=((4,), torch.float32))
        ret[rank + 10] = bool(torch.equal(full, fn(x)))
    finally:
        dist.destroy_process_group()


def test_row_shape_disagreement_is_an_error_not_a_hang():
    """A rank that owns no row of a batch takes the output row shape from ``out_like`` (or the input row): if ``fn``
    changes the shape and the caller forgot ``out_like``, the ranks would all-gather mismatched buffers -- the sharder
    must detect that (one tiny all-gather per batch shape) and raise on every rank."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_mismatch_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    join_all(procs, 120, "shape-mismatch detection over gloo")
    assert ret[0] == "raised" and ret[1] == "raised", dict(ret)
    assert ret[10] and ret[11]


def _forced_worker(port, ret):
    single_thread()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        x = torch.randn(5, 4, 8, 8)
        plain, forced = RowSharder(), RowSharder(force_exchange=True)
        a, b = plain.run(_model, x), forced.run(_model, x)
        ret["ok"] = bool(torch.equal(a, b)) and plain.exchanges == 0 and forced.exchanges == 1 and not plain.exchange
    finally:
        dist.destroy_process_group()


def test_force_exchange_runs_the_collective_on_one_rank():
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_forced_worker, args=(_free_port(), ret))
    p.start()
    join_all([p], 120, "forced exchange on one rank")
    assert ret["ok"]
