"""``tree_model`` - the object main.py drives (reference model/clip_tree.py:19-333), on libhgr.so.

Same constructor, attributes and methods as the reference class, so the
reference's train/eval scripts can construct and call it unchanged:

    tree_model(opts, candidates_train, candidates_test)
    .update_classifier()  .__call__(imgs, targets) -> logits [B, N]  .save(opts, epoch)
    .nodes .c2p .p2c .d2n .start_up .train_index .test_index .resolution .save_path .layer_weight

Differences, all on purpose:
  * logits are fp32 (the reference returns fp16 on GPU), computed by the MFMA GEMM of hgr_gemm_nt;
  * prompts: if ``nltk``'s WordNet is importable the names come from it exactly like the reference
    (clip_tree.py:52-60); otherwise pass ``node_tokens`` (int64 [N, 77]) - there is no WordNet offline;
  * extra device arrays for the fused evaluation (depth per node, int32 index copies).
"""
from __future__ import annotations

import copy
import math
import os
import random
from typing import Optional, Sequence

import torch
import torch.nn as nn

from .. import clip, ops
from ..hierarchy import build_hierarchy
from .._lib import HgrError

TEMPLATE = "a photo of a {}."     # data/templates.py TEMPLATES_SIMPLE[0], the only one used (clip_tree.py:52)

# Evaluation steps as a two-stage pipeline (forward_eval_overlapped): the class-token tail of step i beside the head of step i + 1.
# HGR_TAIL_OVERLAP=0 (or clip_tree.TAIL_OVERLAP = False) keeps every step one graph on one stream.
TAIL_OVERLAP = os.environ.get("HGR_TAIL_OVERLAP", "1") != "0"
# The class operand of the class-logits product at ~22 bits instead of 11, by concatenation along K (round 5; the default of BOTH
# routes - forward() and the fused evaluation - since round 6): "class": [f | f] . [z_hi | z_lo]^T = f . (z_hi + z_lo) removes the 16-bit
# rounding of the class rows from the logits - the rounding that flips top-k / level ids against the fp32 reference (every class row
# is rounded differently; the feature row's rounding moves all of a row's logits together) - at twice the K of an 11 GFLOP product
# (+0.2 % per step).  Embedding widths <= 512 (the evaluation kernel keeps K <= 1 024); wider towers (RN50: 1 024) keep the plain
# 16-bit operand.  HGR_LOGITS_SPLIT=none switches it off, =feat splits the feature operand instead ([f_hi | f_lo] . [z | z]^T: measured,
# gains nothing; profiles/NOTES.md, round 5).
LOGITS_SPLIT = os.environ.get("HGR_LOGITS_SPLIT", "class")


class _StopHead(Exception):
    """Raised by the split hook while the HEAD graph of a pipelined step is captured: the rest of the step belongs to the tail graph."""


class tree_model(nn.Module):
    def __init__(self, opts, candidates_train: Sequence[str], candidates_test: Sequence[str],
                 node_tokens: Optional[torch.Tensor] = None, clip_model: Optional[nn.Module] = None):
        super().__init__()
        self.opts = opts
        self.device = f"cuda:{opts.device}" if isinstance(opts.device, int) else opts.device
        self.save_path = "{}/{}/{}_{}_{}/".format(opts.folder, opts.exp_name, opts.weights, opts.out_ratio, opts.in_ratio)
        self.file_path = self.save_path + "clip_{}".format(opts.from_epoch)
        os.makedirs(self.save_path, exist_ok=True)

        # semantic structure (utils.py:39-72)
        import json
        with open(opts.graph_path, "r") as f:
            self.hierarchy = build_hierarchy(json.load(f))
        self.p2c, self.c2p, self.d2n, self.nodes, self.start_up = self.hierarchy.as_tuple()
        self.nodes_id = list(range(len(self.nodes)))

        # CLIP (clip_tree.py:34-48)
        if clip_model is None:
            clip_model, _ = clip.load(name=opts.arch, device=self.device, download_root="pretrained",
                                      image_dtype=getattr(opts, "image_dtype", "f16"),
                                      text_dtype=getattr(opts, "text_dtype", "f16"))
        self.clip_model = clip_model
        if getattr(opts, "fetch", False):
            self.clip_model.load_state_dict(torch.load(opts.fetch_path, map_location=self.device))
        if getattr(opts, "load", False):
            path = self.file_path if opts.load_path == "none" else opts.load_path
            self.clip_model.load_state_dict(torch.load(path, map_location=self.device))
            print("successfully loaded")
        self.clip_model.eval()
        for p in self.clip_model.parameters():
            p.requires_grad_(True)
        self.loss = nn.CrossEntropyLoss()

        # prompts -> token ids (clip_tree.py:52-60).  With opts.n_ctx > 0 the template is CoOp's learnable context
        # (model/CoOp.py:58-79: "X X ... X {name}." with n_ctx generic context vectors, init N(0, 0.02)); caller-supplied
        # node_tokens must then carry n_ctx placeholder tokens after SOT.
        self.n_ctx = int(getattr(opts, "n_ctx", 0) or 0)
        if node_tokens is None:
            tmpl = (" ".join(["X"] * self.n_ctx) + " {}.") if self.n_ctx else TEMPLATE
            node_tokens = clip.tokenize([tmpl.format(self._wordnet_name(n)) for n in self.nodes])
        self.ctx = None
        if self.n_ctx:
            wt = self.clip_model.transformer.width
            self.ctx = nn.Parameter(torch.empty(self.n_ctx, wt, device=self.device).normal_(std=0.02))
        if node_tokens.shape[0] != len(self.nodes):
            raise ValueError(f"node_tokens has {node_tokens.shape[0]} rows for {len(self.nodes)} nodes")
        self.node_tokens = node_tokens.long().to(self.device)

        # misc (clip_tree.py:63-74)
        self.resolution = self.clip_model.visual.input_resolution
        self.candidates_train, self.candidates_test = candidates_train, candidates_test
        index = {n: i for i, n in enumerate(self.nodes)}
        self.train_index = torch.tensor([index[c] for c in candidates_train]).to(self.device)
        self.test_index = torch.tensor([index[c] for c in candidates_test]).to(self.device)
        self.max_depth = max(self.d2n.keys())
        if opts.weights == "adaptive":
            num_layer = [len(self.d2n[layer]) for layer in self.d2n.keys()]
            # a real leaf parameter (the reference multiplies after wrapping, which makes it a
            # non-parameter and breaks SGD([layer_weight]): SURVEY.md F11-ii)
            # on the model's device: get_weights('adaptive') multiplies it with device-side loss terms (the reference only ever
            # multiplies 0-dim tensors, which torch allows across devices; a [K] vector it does not)
            self.layer_weight = nn.Parameter(((1.0 / torch.tensor(num_layer, dtype=torch.float32)) * opts.scale).to(self.device))

        # device arrays for the evaluation kernels
        self.train_index32 = self.train_index.to(torch.int32)
        self.test_index32 = self.test_index.to(torch.int32)
        self.depth32 = torch.from_numpy(self.hierarchy.depth).to(self.device)
        self.zsl_weights = None
        self._zsl16 = None
        self.use_graph = os.environ.get("HGR_GRAPH", "1") != "0"     # replay forward() as a HIP graph (HGR_GRAPH=0: eager launches)
        self._graphs, self._graph_gen, self._graph_misses, self._graph_static = {}, None, 0, None
        self._pipe = None            # state of the two-stage evaluation pipeline (forward_eval_overlapped)

    @staticmethod
    def _wordnet_name(wnid: str) -> str:
        try:
            from nltk.corpus import wordnet as wn
        except Exception as e:  # noqa: BLE001
            raise RuntimeError("WordNet (nltk) is not available: pass node_tokens= to tree_model") from e
        return wn.synset_from_pos_and_offset("n", int(wnid[1:])).name().split(".")[0].replace("_", " ")

    def save(self, opts, epoch):
        torch.save(self.clip_model.state_dict(), self.save_path + "clip_{}".format(epoch))

    # ---------------------------------------------------------------------------------------------
    # zero-shot forward path
    # ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def update_classifier(self, group=None):
        """Text-encode every node prompt, L2-normalise rows -> ``zsl_weights`` [N, D] fp32
        (clip_tree.py:318-325).  With a process ``group`` each rank encodes N/world rows and the rows
        are all-gathered over RCCL (hgr_net_amd.parallel)."""
        n = len(self.nodes)
        if group is not None:
            from ..parallel import sharded_text_features
            feats = sharded_text_features(self.clip_model, self.node_tokens, group, ctx=self.ctx)
        else:
            feats = self.clip_model.encode_text(self.node_tokens, ctx=self.ctx)
        z32 = torch.empty_like(feats)
        z16 = torch.empty(feats.shape, dtype=self.clip_model.image_dtype, device=feats.device)
        ops.l2norm_rows(feats, y16=z16, y32=z32)
        self.zsl_weights, self._zsl16 = z32, z16
        assert z32.shape[0] == n

    @torch.no_grad()
    def forward(self, inputs, targets=None, static_output: bool = False):
        """logits[B, N] = normalise(encode_image(x)) @ zsl_weights.T, no temperature; ``targets`` is
        ignored as in the reference (clip_tree.py:328-333)."""
        if self._zsl16 is None:
            raise HgrError("call update_classifier() before forward()")
        self.join_tail()            # a pipelined evaluation step may still be reading workspace "v" on the side stream
        if self.use_graph and inputs.is_cuda:
            return self._forward_graphed(inputs, static_output)
        return self._forward_eager(inputs)

    @torch.no_grad()
    def forward_eval(self, inputs, plan, k: int):
        """Evaluation-only forward: (level arg-max [B, L], top-1 [B, 1], top-k [B, k]) int32 node ids of main.py:136-176 from
        the image batch, with the class-logits GEMM and its consumers fused (ops.logits_eval): the [B, N] logits of
        clip_tree.py:331 are never written.  `forward()` keeps returning the logits for callers that want them.
        Outputs are static buffers of a replayed HIP graph: consume them before the next call."""
        if self._zsl16 is None:
            raise HgrError("call update_classifier() before forward_eval()")
        self.join_tail()
        plan.bind(self._eval_class_operand(), self._zsl16.shape[1])
        if self.use_graph and inputs.is_cuda:
            return self._forward_graphed(inputs, True, ("eval", plan, k))
        return self._forward_eager(inputs, ("eval", plan, k))

    def _forward_graphed(self, inputs, static_output: bool = False, mode=None):
        """The ~100 launches of one forward replayed as a HIP graph: no host launch cost and no inter-kernel gaps
        (+6 % on the ViT-B/32 step).  Same kernels, same bits.  A graph is bound to the buffers it was captured on, so
        graphs live for one generation = (input shape, dtype, classifier, prepared weights): anything else clears them
        (the warm-up run may also have re-allocated workspace buffers older graphs point into).  Inside a generation up
        to 4 graphs are keyed by the input buffer's address - loaders recycle a few buffers; after 8 misses in a row the
        input is copied into one static buffer instead.  The logits are returned as a fresh tensor unless the caller passes
        ``static_output=True`` (it consumes them before the next forward: the evaluation loop does)."""
        def generation():
            return (tuple(inputs.shape), inputs.dtype, self._zsl16.data_ptr(), self.clip_model._ws.epoch, self.clip_model._fingerprint(),
                    tuple(sorted(self.clip_model._ln_off)), None if mode is None else (mode[0], id(mode[1]), mode[1].zsl.data_ptr(), mode[2]))

        self.clip_model.poll_ln_guard()          # every 64th call: a tripped range guard switches the tower (and moves the generation below)

        if generation() != self._graph_gen:
            # warm-up BEFORE the key is fixed: it builds the prepared weights and may (re)allocate workspace buffers, both of
            # which move the key; graphs captured afterwards then see a stable generation.  A direct encode_image call with
            # a larger batch between two forwards re-allocates the workspace -> epoch moves -> stale graphs are dropped here.
            self._graphs.clear()
            self._forward_eager(inputs, mode)
            self._graph_gen, self._graph_misses, self._graph_static = generation(), 0, None
        ent = self._graphs.get(inputs.data_ptr())
        if ent is None:
            self._graph_misses += 1
            if self._graph_misses > 8:                          # addresses never repeat: one static input buffer
                if self._graph_static is None:
                    buf = torch.empty_like(inputs)
                    buf.copy_(inputs)
                    self._graph_static = (buf,) + self._capture(buf, mode)
                buf, g, out = self._graph_static
                buf.copy_(inputs)
                g.replay()
                return out if (static_output or mode is not None) else out.clone()
            if len(self._graphs) >= 4:
                self._graphs.pop(next(iter(self._graphs)))
            ent = self._graphs[inputs.data_ptr()] = self._capture(inputs, mode)
        else:
            self._graph_misses = 0
        ent[0].replay()
        return ent[1] if (static_output or mode is not None) else ent[1].clone()

    # ---------------------------------------------------------------------------------------------
    # evaluation steps as a two-stage pipeline
    # ---------------------------------------------------------------------------------------------
    def _eager_phase(self, inputs, mode, phase: str, tag: str):
        """One half of an evaluation step on the current stream, workspace set ``tag``.  'head': every launch up to ops.split_point()
        (patch unfold and GEMM, blocks 0 .. n-2, keys / values of the last block).  'tail': the rest (the last block on the class-token
        rows, visual head, L2 norm, class logits + evaluation); the host code of the head is walked with its launches muted so that the
        tail sees exactly the views and buffers the head wrote.  Returns (outputs or None, whether the split point was reached)."""
        from .. import _lib
        seen = []
        if phase == "head":
            def hook(name):
                seen.append(name)
                raise _StopHead()
        else:
            def hook(name):
                seen.append(name)
                _lib.set_muted(False)
        prev_hook, prev_tag = ops.SPLIT_HOOK, self.clip_model._img_tag
        ops.SPLIT_HOOK, self.clip_model._img_tag = hook, tag
        out = None
        try:
            _lib.set_muted(phase == "tail")
            out = self._forward_eager(inputs, mode)
        except _StopHead:
            pass
        finally:
            _lib.set_muted(False)
            ops.SPLIT_HOOK, self.clip_model._img_tag = prev_hook, prev_tag
        return out, bool(seen)

    def _pipe_state(self, dev):
        if self._pipe is None:
            # the tail's launches are a few dozen workgroups each: a high-priority stream lets them take the first slots that free up
            side = torch.cuda.Stream(device=dev, priority=int(os.environ.get("HGR_TAIL_PRIO", "-1")))
            self._pipe = {"side": side, "graphs": {}, "gen": None, "ok": None, "step": 0, "misses": 0, "static": {},
                          "head_done": [torch.cuda.Event(), torch.cuda.Event()], "tail_done": [torch.cuda.Event(), torch.cuda.Event()]}
            for e in self._pipe["tail_done"]:
                e.record()
        return self._pipe

    def join_tail(self) -> None:
        """Order the current stream behind every tail launched so far (readers of the evaluation counters call this)."""
        if self._pipe is not None:
            torch.cuda.current_stream().wait_stream(self._pipe["side"])

    @torch.no_grad()
    def forward_eval_overlapped(self, inputs, plan, k: int, consume) -> bool:
        """forward_eval as a two-stage pipeline over consecutive calls.  The last image block, the visual head, the class logits and the
        evaluation act on B rows - a few dozen workgroups per launch, ~0.17 ms of a 5.4 ms ViT-B/32 step with the chip idle.  Here
        every step is TWO HIP graphs: the head on the caller's stream, the tail on a second stream behind an event, each step parity on
        its own workspace set, so the tail of step i runs beside the head of step i + 1 (which may not start on a workspace set before
        the tail that last read it has finished).  ``consume(level_ids, top1, topk)`` is called with the second stream current: its
        launches (hgr_eval_counters) are ordered behind the tail and ahead of the next use of those static outputs.  Same kernels on
        the same data as forward_eval: same ids.  Returns False when the step cannot be split (no HIP graphs, a tower without a
        class-token tail): the caller then takes forward_eval."""
        if not (TAIL_OVERLAP and self.use_graph and inputs.is_cuda) or self._zsl16 is None:
            return False
        if inputs.dtype not in (torch.float32, torch.uint8) or not inputs.is_contiguous():
            # the head's host code is walked a second time (launches muted) to capture the tail: an input conversion would be
            # re-executed there and captured into the tail graph - the pipelined route takes the tower's own input forms only
            return False
        from ..clip import model as _clip_model
        if _clip_model.IMG_STREAMS > 1:
            # the sliced image tower (HGR_IMG_STREAMS >= 2) names its workspace sets "v", "v1", ... itself and ignores _img_tag: both
            # step parities would share one set and the tail of step i would race with the head of step i + 1
            return False
        plan.bind(self._eval_class_operand(), self._zsl16.shape[1])
        mode = ("eval", plan, k)
        st = self._pipe_state(inputs.device)
        self.clip_model.poll_ln_guard()
        gen = (tuple(inputs.shape), inputs.dtype, self._zsl16.data_ptr(), self.clip_model._fingerprint(), tuple(sorted(self.clip_model._ln_off)),
               id(plan), plan.zsl.data_ptr(), k)
        main = torch.cuda.current_stream()
        if gen != st["gen"]:
            # new generation (shape, classifier, weights): both workspace sets exist after one eager step each; whether the step has
            # a split point is learnt from the head phase
            main.wait_stream(st["side"])
            st["graphs"].clear()
            st["static"].clear()
            st["misses"] = 0
            ok = True
            for tag in ("v", "v@1"):
                ok &= self._eager_phase(inputs, mode, "head", tag)[1]
                if ok:
                    self._eager_phase(inputs, mode, "tail", tag)
            main.synchronize()
            st["gen"], st["ok"] = gen, ok
        if not st["ok"]:
            return False
        par = st["step"] & 1
        st["step"] += 1
        tag = "v@1" if par else "v"
        key = (inputs.data_ptr(), par, self.clip_model._ws.epoch)
        ent = st["graphs"].get(key)
        if ent is None:
            st["misses"] += 1
            if st["misses"] > 8:                              # input addresses never repeat: one static input buffer per parity
                buf = st["static"].get(par)
                if buf is None:
                    buf = st["static"][par] = torch.empty_like(inputs)
                buf.copy_(inputs)                             # on the caller's stream, behind the head that last read it
                inputs = buf
                key = (buf.data_ptr(), par, self.clip_model._ws.epoch)
                ent = st["graphs"].get(key)
        else:
            st["misses"] = 0
        if ent is None:
            if len(st["graphs"]) >= 8:
                main.wait_stream(st["side"])
                st["graphs"].clear()
            main.wait_stream(st["side"])
            main.synchronize()
            gh, gt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(gh):
                self._eager_phase(inputs, mode, "head", tag)
            with torch.cuda.graph(gt):
                outs, _ = self._eager_phase(inputs, mode, "tail", tag)
            if key[2] != self.clip_model._ws.epoch:          # a capture re-allocated a workspace buffer: the warm-up above should have made that impossible
                raise HgrError("workspace buffers moved during the capture of a pipelined evaluation step")
            ent = st["graphs"][key] = (gh, gt, outs)
        gh, gt, outs = ent
        main.wait_event(st["tail_done"][par])               # the tail that last read this workspace set (step i - 2)
        gh.replay()
        st["head_done"][par].record(main)
        side = st["side"]
        with torch.cuda.stream(side):
            side.wait_event(st["head_done"][par])
            gt.replay()
            consume(*outs)
            st["tail_done"][par].record(side)
        return True

    def _capture(self, inputs, mode=None):
        self._forward_eager(inputs, mode)                       # warm-up: workspace buffers and prepared weights exist
        torch.cuda.current_stream().synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self._forward_eager(inputs, mode)
        return g, out

    def _eval_class_operand(self):
        """The 16-bit class matrix hgr_logits_eval multiplies with: `_zsl16`, or with HGR_LOGITS_SPLIT its K-concatenated form."""
        split = LOGITS_SPLIT if (LOGITS_SPLIT in ("class", "feat") and 2 * self._zsl16.shape[1] <= 1024) else ""
        if not split:
            return self._zsl16
        key = (split, self._zsl16.data_ptr(), self._zsl16._version, self.zsl_weights.data_ptr(), self.zsl_weights._version)
        if getattr(self, "_zsl_split_key", None) != key:
            z16 = self._zsl16
            second = (self.zsl_weights.float() - z16.float()).to(z16.dtype) if split == "class" else z16
            self._zsl_split, self._zsl_split_key = torch.cat([z16, second], dim=1).contiguous(), key
        return self._zsl_split

    def _forward_eager(self, inputs, mode=None):
        feats = self.clip_model.encode_image(inputs)
        b, n = feats.shape[0], self._zsl16.shape[0]
        f16 = torch.empty(feats.shape, dtype=self._zsl16.dtype, device=feats.device)
        zc = self._eval_class_operand() if mode is None else mode[1].zsl       # both routes multiply with the same class operand
        if zc is not None and zc.shape[1] == 2 * feats.shape[1]:
            # LOGITS_SPLIT: the features at twice the width to match the K-concatenated class matrix
            if LOGITS_SPLIT == "feat":
                f32 = torch.empty_like(feats)
                ops.l2norm_rows(feats, y16=f16, y32=f32)
                second = (f32 - f16.float()).to(f16.dtype)
            else:
                ops.l2norm_rows(feats, y16=f16)
                second = f16
            fin = torch.cat([f16, second], dim=1)
        else:
            ops.l2norm_rows(feats, y16=f16)
            fin, zc = f16, (self._zsl16 if mode is None else zc)
        if mode is not None:                                    # ("eval", plan, k): logits GEMM + evaluation fused, no logits written
            return ops.logits_eval(fin, mode[1], mode[2])
        ld = (n + 63) // 64 * 64                   # 16-byte aligned rows for the vector stores
        logits = torch.empty((b, ld), dtype=torch.float32, device=feats.device)
        ops.gemm_nt(fin, zc, logits, n=n, tag="logits")
        return logits[:, :n]

    # ---------------------------------------------------------------------------------------------
    # host-side sampling / weighting logic of the training step (no tensors involved)
    # ---------------------------------------------------------------------------------------------
    def get_contra_ids(self, method, target, depth=None, parents=None):
        """Negative-class candidates + position of the target (clip_tree.py:80-196: 'random', 'topk',
        'near_simi', 'brothers').  Returns (list of node ids with the target inside, index of the target)."""
        nc = self.opts.num_compare
        if method == "random":
            ids = random.sample(self.train_index.tolist(), nc)
        elif method == "topk":
            low, high = min(self.d2n.keys()), max(self.d2n.keys())
            if depth - self.opts.k > low:
                low = depth - self.opts.k
            cand = []
            for d in range(low, depth):
                cand.extend(self.d2n[d])
            if depth == 0:
                cand.extend(self.d2n[depth])
            ids = list(set(cand) - set(parents))
            if len(ids) > nc:
                ids = random.sample(ids, nc)
        elif method == "brothers":
            if len(parents) > 1 and depth > 0:
                ids = copy.copy(self.p2c[parents[depth - 1]])
            else:
                ids = copy.copy(self.start_up)
            if len(ids) > nc:
                ids = random.sample(ids, nc)
        elif method == "near_simi":
            # clip_tree.py:143-178: the num_compare prompts most similar to the target's (text cosine under the CURRENT
            # weights) among the nodes within k levels, ancestors and children excluded.  The reference's version slices
            # the wrong axis (`argsort(...)[:num_compare]` on a [1, n] tensor) and dies building a ragged tensor; this is
            # its evident intent.  Candidates are walked in ascending node id (the reference iterates a set of ints).
            low, high = min(self.d2n.keys()), max(self.d2n.keys())
            low = max(low, depth - self.opts.k)
            high = min(high, depth + self.opts.k)
            cand = set()
            for d in range(low, high + 1):
                cand.update(self.d2n[d])
            cand = sorted(cand - set(parents) - set(self.p2c[target]) - {target})
            nc = min(nc, len(cand))
            with torch.no_grad():
                toks = self.node_tokens[torch.tensor([target] + cand, device=self.node_tokens.device)]
                f = self.clip_model.encode_text(toks, ctx=self.ctx.data if self.ctx is not None else None).float()
                f = f / f.norm(dim=-1, keepdim=True)
                order = (f[1:] @ f[0]).argsort(descending=True, stable=True)[:nc].tolist()
            ids = [cand[i] for i in order]
        else:
            raise NotImplementedError(f"sample_strategy {method!r} (the reference's 'simi' reads attributes that do not exist)")
        if target not in ids:
            ids.append(target)
        return ids, ids.index(target)

    def get_weights(self, method, max_depth=None, device=None):
        """Layer weights (clip_tree.py:198-219).  ``device`` (not in the reference): where the closed-form weights are made - the
        training step asks for "cpu" (it only needs their float values; a device tensor per inner step is a synchronous
        host-to-device copy plus a read-back, i.e. two drains of the launch queue)."""
        dev = self.device if device is None else device
        if method == "equal":
            return (torch.ones(max_depth) / max_depth).to(dev)
        if method == "decreasing":
            w = torch.arange(start=max_depth, end=0, step=-1).to(dev)
        elif method == "increasing":
            w = torch.arange(start=1, end=max_depth + 1).to(dev)
        elif method == "nl_increasing":
            w = (torch.arange(start=1, end=max_depth + 1) ** 3).to(dev)
        elif method == "nl_decreasing":
            w = (torch.arange(start=max_depth, end=0, step=-1) ** 3).to(dev)
        elif method == "adaptive":
            return torch.softmax(100 ** self.layer_weight[:max_depth], dim=0)
        else:
            raise ValueError(method)
        return w / w.sum()

    def outer_inner_plan(self, target: int):
        """The (p_out, p_in, depth, parents_in, k_loop, m_loop, K, M) schedule of one OM step
        (clip_tree.py:228-251)."""
        parents = copy.copy(self.c2p[target]) + [target]
        k = max(1, math.ceil(self.opts.out_ratio * len(parents)))
        outer = parents[::-1][:k]
        plan = []
        for k_loop, p_out in enumerate(outer):
            parents_in = copy.copy(self.c2p[p_out]) + [p_out]
            m = max(1, math.ceil(self.opts.in_ratio * len(parents_in)))
            inner = parents_in[::-1][:m]
            for m_loop, p_in in enumerate(inner):
                plan.append(dict(p_out=p_out, p_in=p_in, depth=parents_in.index(p_in), parents_in=parents_in,
                                 k_loop=k_loop, m_loop=m_loop, K=len(outer), M=len(inner)))
        return plan

    def train_batch(self, inputs, targets, training_method, sample_strategy):
        """One training step (clip_tree.py:222-316): returns the summed loss as a float and leaves the
        gradients ACCUMULATED in ``.grad`` of every CLIP parameter, like the reference's autograd calls.
        Runs on libhgr (hgr_net_amd.training); compute dtype ``opts.train_dtype`` (default bf16)."""
        if getattr(self, "_trainer", None) is None:
            from ..training import OMTrainer
            self._trainer = OMTrainer(self, getattr(self.opts, "train_dtype", "bf16"))
        return self._trainer.train_batch(inputs, targets, training_method, sample_strategy)
