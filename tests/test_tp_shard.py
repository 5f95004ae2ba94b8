"""TP-aware weight update: every tensor-parallel rank of an inference engine receives only the
slices it stores, where the reference sends each full tensor to every rank (vllm1.py:110-127).

CPU tests: the cut rules on the Qwen2.5-7B parameter list, bucket plans that trainer and workers derive
independently, the whole request -> receive path through `WeightUpdateManager` and `WorkerExtension` in one
process, and five gloo processes (one trainer, two TP = 2 engines) with one subgroup per TP rank."""

import json
import os
import socket
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

from pipelinerl_amd.finetune_loop import WeightUpdateManager, WeightUpdateRequest
from pipelinerl_amd.tp_shard import TpShard, default_tp_rule, plan_tp_shards, shard_view
from pipelinerl_amd.vllm_worker import StandaloneShardReceiver, WorkerExtension
from pipelinerl_amd.weight_sync import ParamSpec, ShardedSender, plan_shard_buckets


def qwen_shapes(layers=2, hidden=3584, inter=18944, heads=28, kv_heads=4, head_dim=128, vocab=152064, tied=False):
    out = [("model.embed_tokens.weight", (vocab, hidden))]
    for i in range(layers):
        p = f"model.layers.{i}."
        out += [(p + "self_attn.q_proj.weight", (heads * head_dim, hidden)), (p + "self_attn.q_proj.bias", (heads * head_dim,)),
                (p + "self_attn.k_proj.weight", (kv_heads * head_dim, hidden)), (p + "self_attn.k_proj.bias", (kv_heads * head_dim,)),
                (p + "self_attn.v_proj.weight", (kv_heads * head_dim, hidden)), (p + "self_attn.v_proj.bias", (kv_heads * head_dim,)),
                (p + "self_attn.o_proj.weight", (hidden, heads * head_dim)),
                (p + "mlp.gate_proj.weight", (inter, hidden)), (p + "mlp.up_proj.weight", (inter, hidden)),
                (p + "mlp.down_proj.weight", (hidden, inter)),
                (p + "input_layernorm.weight", (hidden,)), (p + "post_attention_layernorm.weight", (hidden,))]
    out.append(("model.norm.weight", (hidden,)))
    if not tied:
        out.append(("lm_head.weight", (vocab, hidden)))
    return out


def test_cut_rules_for_the_qwen_family():
    cuts = plan_tp_shards(qwen_shapes(), tp_size=2)
    L = "model.layers.1."
    assert cuts["model.embed_tokens.weight"] == TpShard(0, 2) and cuts["lm_head.weight"] == TpShard(0, 2)
    for n in ("q_proj", "k_proj", "v_proj"):
        assert cuts[L + f"self_attn.{n}.weight"] == TpShard(0, 2) and cuts[L + f"self_attn.{n}.bias"] == TpShard(0, 2)
    assert cuts[L + "mlp.gate_proj.weight"] == TpShard(0, 2) and cuts[L + "mlp.up_proj.weight"] == TpShard(0, 2)
    assert cuts[L + "self_attn.o_proj.weight"] == TpShard(1, 2) and cuts[L + "mlp.down_proj.weight"] == TpShard(1, 2)
    assert cuts[L + "input_layernorm.weight"] == TpShard() and cuts["model.norm.weight"] == TpShard()
    # shapes a TP = 2 engine stores
    assert cuts[L + "self_attn.q_proj.weight"].shard_shape((3584, 3584)) == (1792, 3584)
    assert cuts[L + "self_attn.o_proj.weight"].shard_shape((3584, 3584)) == (3584, 1792)
    assert cuts[L + "self_attn.k_proj.weight"].bounds((512, 3584), 1, 2) == (256, 256)
    # TP 1: nothing is cut
    assert set(plan_tp_shards(qwen_shapes(), 1).values()) == {TpShard()}


def test_fewer_kv_heads_than_tp_ranks_replicates_in_groups():
    """4 KV heads on 8 TP ranks: k / v are cut into 4 pieces, ranks 2h and 2h + 1 both hold head h (vLLM's
    num_kv_head_replicas); without the head count the rule must not guess a cut along a head boundary it cannot see."""
    name = "model.layers.0.self_attn.k_proj.weight"
    cut = default_tp_rule(name, (512, 3584), tp_size=8, kv_heads=4)
    assert cut == TpShard(0, 4)
    assert [cut.piece(t, 8) for t in range(8)] == [0, 0, 1, 1, 2, 2, 3, 3]
    assert cut.bounds((512, 3584), 5, 8) == (256, 128)
    assert default_tp_rule("model.layers.0.self_attn.q_proj.weight", (3584, 3584), 8, kv_heads=4) == TpShard(0, 8)
    # 3 KV heads on 8 ranks cannot be grouped evenly; a dimension that does not divide stays whole
    assert default_tp_rule(name, (384, 3584), 8, kv_heads=3) == TpShard()
    assert default_tp_rule("lm_head.weight", (151937, 896), 2) == TpShard()
    # overrides win
    assert plan_tp_shards([(name, (512, 3584))], 2, overrides={name: TpShard(1, 2)})[name] == TpShard(1, 2)


def test_slices_reassemble_and_plans_agree():
    torch.manual_seed(0)
    shapes = qwen_shapes(layers=1, hidden=64, inter=160, heads=4, kv_heads=2, head_dim=16, vocab=96)
    full = {n: torch.randn(s) for n, s in shapes}
    for tp in (2, 4):
        cuts = plan_tp_shards(shapes, tp, kv_heads=2)
        for n, t in full.items():
            c = cuts[n]
            if c.dim is None:
                assert shard_view(t, c, 1, tp) is t
                continue
            pieces = [shard_view(t, c, r, tp) for r in range(tp)]
            uniq = [pieces[r] for r in range(tp) if r == 0 or c.piece(r, tp) != c.piece(r - 1, tp)]
            assert torch.equal(torch.cat(uniq, dim=c.dim), t)
        specs = [ParamSpec(n, tuple(s), torch.float32) for n, s in shapes]
        sizes = []
        for r in range(tp):
            groups, mine = plan_shard_buckets(specs, cuts, r, tp, bucket_bytes=4096)
            assert len(groups) == len(mine) and [[sp.name for sp, _ in g] for g in groups] == [[sp.name for sp, _ in b] for b in mine]
            assert all(off % 256 == 0 for b in mine for _, off in b)
            for b in mine:
                for sp, _ in b:
                    assert sp.shape == cuts[sp.name].shard_shape(dict(shapes)[sp.name])
            sizes.append(sum(sp.nbytes for b in mine for sp, _ in b))
        assert len(set(sizes)) == 1  # every rank gets the same number of bytes
        total = sum(sp.nbytes for sp in specs)
        replicated = sum(sp.nbytes for sp in specs if cuts[sp.name].dim is None)
        kv_extra = sum(sp.nbytes * (1 / cuts[sp.name].parts - 1 / tp) for sp in specs if cuts[sp.name].dim is not None)
        assert sizes[0] == pytest.approx((total - replicated) / tp + replicated + kv_extra)


class LoopGroups:
    """The per-TP-rank communicators in one process: what the trainer broadcasts on group t is replayed to the
    workers of TP rank t."""

    class One:
        device = torch.device("cpu")

        def __init__(self):
            self.sent, self.cursor = [], {}

        def broadcast_bucket(self, buf, mode="scatter_allgather", src=0):
            self.sent.append(buf.clone())

        def reader(self, who):
            outer = self

            class R:
                device = torch.device("cpu")

                def broadcast_bucket(self, buf, mode="scatter_allgather", src=0):
                    k = outer.cursor.get(who, 0)
                    buf.copy_(outer.sent[k])
                    outer.cursor[who] = k + 1

                def close(self):
                    pass

            return R()

    def __init__(self, tp):
        self.groups = [LoopGroups.One() for _ in range(tp)]


def test_manager_to_workers_in_one_process():
    """`WeightUpdateManager(transport="sharded")` -> request -> two TP = 2 engines: every worker ends up with exactly
    `narrow(dim, ...)` of the trainer's tensors, got S / 2 (+ norms) bytes, and the request states the cut."""
    torch.manual_seed(1)
    shapes = qwen_shapes(layers=2, hidden=64, inter=160, heads=4, kv_heads=2, head_dim=16, vocab=96)

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ps = torch.nn.ParameterDict({n.replace(".", "__"): torch.nn.Parameter(torch.randn(s).to(torch.bfloat16)) for n, s in shapes})

        def named_parameters(self, *a, **k):  # the trainer's names
            return [(n.replace("__", "."), p) for n, p in self.ps.items()]

    model = Model()
    loops = LoopGroups(2)
    requests = []
    mgr = WeightUpdateManager(["http://e0", "http://e1"], model, None, loops.groups, transport="sharded", bucket_bytes=8192,
                              post=lambda url, payload: requests.append((url, payload)), kv_heads=2)
    mgr.send_weight_update(7)
    mgr.shutdown()
    assert len(requests) == 2 and all(r[1]["transport"] == "sharded" and r[1]["tp_size"] == 2 for r in requests)
    info = {i["name"]: i for i in requests[0][1]["parameters_info"]}
    assert info["model.layers.0.self_attn.o_proj.weight"]["shard_dim"] == 1 and "shard_dim" not in info["model.norm.weight"]
    assert info["model.layers.0.self_attn.q_proj.weight"]["shape"] == [64, 64]  # full shape, as in the reference message
    full = dict(model.named_parameters())
    total = sum(p.numel() * 2 for p in full.values())
    for engine in range(2):
        for tp_rank in range(2):
            w = StandaloneShardReceiver(shapes, lambda n: torch.bfloat16, torch.device("cpu"), tp_rank, 2, kv_heads=2)
            w.model_update_group = loops.groups[tp_rank].reader((engine, tp_rank))
            w.tp_rank, w.tp_size = tp_rank, 2
            w.receive_weight_update(json.dumps(requests[engine][1]))
            for n, t in full.items():
                assert torch.equal(w.slices[n], shard_view(t.detach(), w.cuts[n], tp_rank, 2)), n
    assert all(0.5 * total <= b <= 0.56 * total for b in mgr._sender.bytes_sent)  # half the parameter set + norms + alignment

    # a worker that was not set up for slices, or runs another TP degree, refuses the update
    class Plain(WorkerExtension):
        device, rank = torch.device("cpu"), 0

    with pytest.raises(RuntimeError):
        Plain().receive_weight_update(json.dumps(requests[0][1]))
    w = StandaloneShardReceiver(shapes, lambda n: torch.bfloat16, torch.device("cpu"), 0, 4)
    w.tp_rank, w.tp_size = 0, 4
    with pytest.raises(ValueError):
        w.receive_weight_update(json.dumps(requests[0][1]))
    # and one without a loader for slices says so
    p = Plain()
    p.tp_rank, p.tp_size, p.model_update_group = 0, 2, loops.groups[0].reader("x")
    with pytest.raises(NotImplementedError):
        p.receive_weight_update(json.dumps(requests[0][1]))


class GlooSubGroup:
    """A per-TP-rank communicator over a gloo subgroup (stands in for WeightSyncGroup.tp_shard_groups on CPU)."""

    device = torch.device("cpu")

    def __init__(self, group, src_global_rank=0):
        self.group, self.src = group, src_global_rank

    def broadcast_bucket(self, buf, mode="scatter_allgather", src=0):
        import torch.distributed as dist

        dist.broadcast(buf, src=self.src, group=self.group)

    def close(self):
        pass


def _rank_main(rank, world, port, out_dir):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tp = 2
    # every process creates every subgroup (torch.distributed rule); subgroup t = trainer + global ranks with (r - 1) % tp == t
    subs = [dist.new_group([0] + [r for r in range(1, world) if (r - 1) % tp == t]) for t in range(tp)]
    shapes = qwen_shapes(layers=2, hidden=64, inter=160, heads=4, kv_heads=2, head_dim=16, vocab=96)
    torch.manual_seed(3)
    full = [(n, torch.randn(s).to(torch.bfloat16)) for n, s in shapes]  # same on every process: the workers use it to check
    cuts = plan_tp_shards(shapes, tp, kv_heads=2)
    if rank == 0:
        sender = ShardedSender([GlooSubGroup(g) for g in subs], bucket_bytes=8192)
        sender.send(full, cuts)
        sender.send(full, cuts)  # a second update reuses the staging buffers
    else:
        t = (rank - 1) % tp
        w = StandaloneShardReceiver(shapes, lambda n: torch.bfloat16, torch.device("cpu"), t, tp, kv_heads=2)
        w.model_update_group = GlooSubGroup(subs[t])
        w.tp_rank, w.tp_size = t, tp
        from pipelinerl_amd.finetune_loop import ParameterInfo

        req = WeightUpdateRequest(version=1, transport="sharded", bucket_bytes=8192, tp_size=tp,
                                  parameters_info=[ParameterInfo(name=n, shape=list(s), dtype="torch.bfloat16", shard_dim=cuts[n].dim,
                                                                 shard_parts=cuts[n].parts) for n, s in shapes])
        ok = True
        for _ in range(2):
            for v in w.slices.values():
                v.zero_()
            w.receive_weight_update(req.model_dump_json())
            ok = ok and all(torch.equal(w.slices[n], shard_view(x, cuts[n], t, tp)) for n, x in full)
        Path(out_dir, f"rank{rank}.json").write_text(json.dumps({"ok": bool(ok), "tp_rank": t}))
    dist.barrier()
    dist.destroy_process_group()


def test_five_processes_two_engines_of_two_tp_ranks(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_main, args=(5, port, str(tmp_path)), nprocs=5, join=True)
    res = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(1, 5)]
    assert all(r["ok"] for r in res) and [r["tp_rank"] for r in res] == [0, 1, 0, 1]


def _vllm_style_rank_storage(full: dict, tp_rank: int, tp: int, heads: int, kv_heads: int, head_dim: int):
    """What vLLM's `load_weights` leaves on TP rank `tp_rank` when it is handed the FULL tensors (the reference path,
    vllm1.py:110-127) - written from the head arithmetic of QKVParallelLinear / MergedColumnParallelLinear /
    RowParallelLinear / VocabParallelEmbedding, without `tp_shard`."""
    out = {}
    q_per = heads // tp
    kv_per = max(1, kv_heads // tp)
    kv_rep = max(1, tp // kv_heads)            # ranks that share one KV head
    kv_first = (tp_rank // kv_rep) * kv_per    # first KV head of this rank
    for name, t in full.items():
        if ".q_proj." in name or ".k_proj." in name or ".v_proj." in name:
            if ".q_proj." not in name:
                continue
            rows = []
            for proj, first, n in (("q_proj", tp_rank * q_per, q_per), ("k_proj", kv_first, kv_per), ("v_proj", kv_first, kv_per)):
                x = full[name.replace("q_proj", proj)]
                rows.append(x[first * head_dim:(first + n) * head_dim])
            out[name.replace("q_proj", "qkv_proj")] = torch.cat(rows, 0)
        elif ".gate_proj." in name:
            up = full[name.replace("gate_proj", "up_proj")]
            n = t.shape[0] // tp
            out[name.replace("gate_proj", "gate_up_proj")] = torch.cat([t[tp_rank * n:(tp_rank + 1) * n], up[tp_rank * n:(tp_rank + 1) * n]], 0)
        elif ".up_proj." in name:
            continue
        elif name.endswith("o_proj.weight") or name.endswith("down_proj.weight"):
            n = t.shape[1] // tp
            out[name] = t[:, tp_rank * n:(tp_rank + 1) * n]
        elif name.endswith("embed_tokens.weight") or name.endswith("lm_head.weight"):
            n = t.shape[0] // tp
            out[name] = t[tp_rank * n:(tp_rank + 1) * n]
        else:
            out[name] = t
    return out


@pytest.mark.parametrize("tp,heads,kv_heads", [(2, 4, 2), (4, 8, 2), (4, 4, 4)], ids=["tp2", "tp4_kv2_replicated", "tp4_kv4"])
def test_sharded_update_lands_in_the_engines_stacked_storage(tp, heads, kv_heads):
    """The engine's REAL layout (round-2 review item 7a): per-rank stacked `qkv_proj` [(q + 2 kv) / tp, H] and
    `gate_up_proj`, including the grouped-query case kv_heads < tp.  `WeightUpdateManager(transport="sharded")` ->
    request -> `StackedShardReceiver`: every byte of the engine's storage equals what vLLM's `load_weights` would have
    produced from the full tensors the reference broadcasts."""
    from pipelinerl_amd.vllm_worker import StackedShardReceiver

    torch.manual_seed(tp * 10 + kv_heads)
    head_dim, hidden, inter = 16, 64, 160
    shapes = qwen_shapes(layers=2, hidden=hidden, inter=inter, heads=heads, kv_heads=kv_heads, head_dim=head_dim, vocab=96)

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ps = torch.nn.ParameterDict({n.replace(".", "__"): torch.nn.Parameter(torch.randn(s).to(torch.bfloat16)) for n, s in shapes})

        def named_parameters(self, *a, **k):
            return [(n.replace("__", "."), p) for n, p in self.ps.items()]

    model = Model()
    loops = LoopGroups(tp)
    requests = []
    mgr = WeightUpdateManager(["http://e0"], model, None, loops.groups, transport="sharded", bucket_bytes=4096,
                              post=lambda url, payload: requests.append((url, payload)), kv_heads=kv_heads)
    mgr.send_weight_update(3)
    mgr.shutdown()
    full = {n: p.detach() for n, p in model.named_parameters()}
    received = 0
    for tp_rank in range(tp):
        w = StackedShardReceiver(shapes, lambda n: torch.bfloat16, torch.device("cpu"), tp_rank, tp, kv_heads=kv_heads)
        q_rows, kv_rows = heads * head_dim // tp, kv_heads * head_dim // min(tp, kv_heads)
        assert tuple(w.storage["model.layers.0.self_attn.qkv_proj.weight"].shape) == (q_rows + 2 * kv_rows, hidden)
        assert tuple(w.storage["model.layers.1.mlp.gate_up_proj.weight"].shape) == (2 * inter // tp, hidden)
        assert not any(".q_proj." in n or ".gate_proj." in n for n in w.storage)
        w.model_update_group = loops.groups[tp_rank].reader((0, tp_rank))
        w.tp_rank, w.tp_size = tp_rank, tp
        for t in w.storage.values():
            t.fill_(7.0)  # every byte must be overwritten
        w.receive_weight_update(json.dumps(requests[0][1]))
        want = _vllm_style_rank_storage(full, tp_rank, tp, heads, kv_heads, head_dim)
        assert set(want) == set(w.storage)
        for n, t in want.items():
            assert torch.equal(w.storage[n], t), (tp_rank, n)
        received += sum(t.numel() * 2 for t in w.storage.values())
    # the TP ranks together received the parameter set once (+ the replicated tensors tp times), not tp times
    total = sum(p.numel() * 2 for p in full.values())
    assert received < 1.35 * total if kv_heads >= tp else received < 1.6 * total
