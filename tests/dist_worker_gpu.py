"""Rank program of tests/test_gpu_multi.py (started with `python -m torch.distributed.run`): data-parallel training step of
the flow decoder over RCCL.  Every rank runs inverse stack (batch-stat BN) + PointFlowNLL + backward on ITS shard of
clouds through the HIP training kernels, then the ONE collective of the step, allreduce_flat_gradients (SURVEY 8e;
lib/networks/training.py:55-56).  Checked on every rank: the reduced flat gradient is the mean of the ranks' local
gradients; on rank 0 additionally: each rank's local gradient equals what a single process computes for that shard
(parity is per replica: BatchNorm statistics are local)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dpf_nets_amd import distributed as D                      # noqa: E402
from dpf_nets_amd import networks as nets, synthetic as SY     # noqa: E402


def local_step(dec, store, p, g):
    store.zero_grad()
    ps, mus, lvs = dec(p, g, mode="inverse")
    pm, pl = torch.zeros_like(p), torch.full_like(p, -3.6)
    loss = nets.PointFlowNLL()(ps + [p], [pm] + mus, [pl] + lvs)
    loss.backward()
    return float(loss)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    # DPF_TEST_BACKEND=gloo + DPF_TEST_SHARE_GPU=1: the ranks are real processes that SHARE cuda:0 and exchange through gloo --
    # how a 1-GPU box runs the 2-rank path (RCCL refuses two ranks on one device); everything but the transport is the same
    backend = os.environ.get("DPF_TEST_BACKEND", "nccl")
    if os.environ.get("DPF_TEST_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    try:
        B, N, G, nf = 4 * world, 256, 128, 2
        state = SY.make_decoder_state(9, nf, 64, G)
        tgt, _, g = SY.synthetic_inputs(9, B, N, G)
        allp, allg = torch.from_numpy(tgt).to(dev), torch.from_numpy(g).to(dev)

        def fresh():
            dec = nets.LocalCondRNVPDecoder(nf, 64, G)
            dec.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()}, strict=True)
            dec = dec.to(dev).train()
            return dec, dec.flatten_parameters()
        dec, store = fresh()
        p, gg = D.shard(allp, allg)                              # this rank's clouds
        assert p.shape[0] == B // world
        local_step(dec, store, p, gg)
        mine = store.flat_g.clone()
        n = D.allreduce_flat_gradients(store)                   # the ONE collective
        assert n == (store.flat_g.numel() if world > 1 else 0)
        locs = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(locs, mine)
        want = sum(locs) / world
        assert torch.allclose(store.flat_g, want, rtol=1e-6, atol=1e-9), float((store.flat_g - want).abs().max())
        if world > 1:
            assert not torch.equal(locs[0], locs[1])             # the shards really differ
        if rank == 0:                                            # per-shard parity against a single process
            for r in range(world):
                lo, hi = D.shard_bounds(B, r, world)
                d2, s2 = fresh()
                local_step(d2, s2, allp[lo:hi].contiguous(), allg[lo:hi].contiguous())
                assert torch.allclose(s2.flat_g, locs[r], rtol=1e-5, atol=1e-8), (r, float((s2.flat_g - locs[r]).abs().max()))
        # eval path shards with no collective at all; per-cloud results come back in batch order
        dec.eval()
        with torch.no_grad():
            ps, _, _ = dec(p, gg, mode="direct")
        per_cloud = ps[-1].square().mean((1, 2))
        gathered = D.gather_clouds(per_cloud)
        assert gathered.shape[0] == B
        # pairwise_CD sharded by rows ("shards by rows with an all-gather at the end"): every rank computes its block of
        # the (N1, N2) matrix, the gathered matrix equals the one a single process computes, bit for bit
        from dpf_nets_amd.networks.utils import pairwise_CD
        gen = torch.Generator().manual_seed(4)
        c1 = (torch.rand(5, 300, 3, generator=gen) - 0.5).to(dev)
        c2 = (torch.rand(7, 257, 3, generator=gen) - 0.5).to(dev)
        full = pairwise_CD(c1, c2)
        mine_rows = pairwise_CD(c1, c2, shard_rows=True)
        assert mine_rows.shape == (5, 7) and torch.equal(mine_rows, full)
        # ---- the whole autoencoder (encoder + posterior + prior flow + base net + decoder, every block on its HIP training
        # kernels): all gradients in ONE flat buffer, the step's exchange is exactly ONE all_reduce -- counted
        from oracle import model_oracle as MO                   # (test infrastructure: the small model configuration only)
        cfg = dict(MO.CONFIG, util_mode="training", g_posterior_n_layers=1, p_decoder_base_type="freevar", p_prior_n_layers=1)
        torch.manual_seed(3)
        model = nets.Local_Cond_RNVP_MC_Global_RNVP_VAE(**cfg).to(dev).train()
        model.flatten_parameters()
        loss_fn = nets.Local_Cond_RNVP_MC_Global_RNVP_VAE_Loss(**cfg)
        arena = D.GradArena(model.parameters())
        assert len(arena.stores) == 2 and arena.n_params == sum(q.numel() for q in model.parameters())
        x = (torch.randn(4, 3, 512, generator=torch.Generator().manual_seed(50 + rank)) * 0.25).to(dev)
        for step in range(3):
            arena.zero_grad()
            torch.manual_seed(7 + step)
            loss = loss_fn(x, x, model(x, x))[0]
            loss.backward()
            arena.sync()
            mine = arena.buf.clone()
            locs = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(locs, mine)
            with D.count_collectives() as cc:
                n = arena.allreduce()
            if world > 1:
                assert cc.total() == 1 and cc.calls == {"all_reduce": 1}, cc.calls
                assert n == arena.buf.numel()
            else:
                assert cc.total() == 0 and n == 0
            want = sum(locs) / world
            assert torch.allclose(arena.buf, want, rtol=1e-6, atol=1e-9), float((arena.buf - want).abs().max())
            assert bool(mine.abs().sum() > 0) and torch.isfinite(mine).all()
        dist.barrier()
        if rank == 0:
            print("DIST_OK world=%d backend=%s flat=%d arena=%d" % (world, backend, store.flat_g.numel(), arena.buf.numel()))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
