"""One GAN training iteration on the HIP engines: the role of train_eval/train_gan.py:train_iter_gan (:13-103).

Same order of operations and the same losses as the reference; what changes is the schedule:
  * the generator forwards of one iteration (G fwd #1 for the D step, #2 for the G step, #3 with shuffled speakers)
    share weights and inputs, so they run as ONE stacked forward with per-call BatchNorm statistics and independent
    dropout / reparameterisation draws; only call #2 is taped and back-propagated (the other two are detached in the
    reference, train_gan.py:39,69);
  * D(real) and D(fake) of the discriminator step run as one stacked forward/backward;
  * the discriminator gradients produced by the generator step are never formed (the reference discards them);
  * nothing reads a loss back to the host inside the iteration; the five .item() calls of :94-102 become one
    deferred read (StepLosses.to_dict()).
  * launches that wait for nothing (gradient zeroing, weight-operand refreshes, the discriminator's dropout draws) ride on the generator
    forward's forked audio branch; the discriminator's head, loss terms and head backward are one launch per pass.
With static shapes the whole iteration is captured into a hipGraph (GraphedGanStep).
"""
import math
import os

import torch

from . import layers as L
from . import ops
from .optim import FusedAdam


class StepLosses:
    """Device-resident loss scalars of one iteration; to_dict() reproduces the reference's return value."""

    def __init__(self, g_scalars, d_scalar, hp, post_warmup):
        self.g, self.d, self.hp, self.post = g_scalars, d_scalar, hp, post_warmup

    def to_dict(self):
        g = self.g.tolist()       # one device->host read
        ops.check_async_errors()  # raises if a persistent kernel's bounded spin timed out (results invalid)
        hp = self.hp
        ret = {"loss": hp["loss_regression_weight"] * g[0]}
        if g[1]:
            ret["KLD"] = hp["loss_kld_weight"] * g[1]
        if g[2]:
            ret["DIV_REG"] = hp["loss_reg_weight"] * g[2]
        if self.post:
            ret["gen"] = hp["loss_gan_weight"] * g[3]
            d = self.d.tolist()       # dis_error itself (separate loss kernel), or the 2 B per-clip log terms of the fused discriminator head
            ret["dis"] = float(d[0]) if len(d) == 1 else -math.fsum(d) / (len(d) // 2)
        return ret


def hyper_params(args):
    keys = ("n_pre_poses", "loss_warmup", "loss_gan_weight", "loss_regression_weight", "loss_kld_weight", "loss_reg_weight",
            "learning_rate", "discriminator_lr_weight")
    return {k: getattr(args, k) for k in keys}


def _stack_inject(inject, tags, prefix):
    """Per-call injected draws ('g1.x', 'g2.x', ...) -> stacked draws ('<prefix>.x') in call order."""
    if inject is None:
        return None
    out = {}
    suffixes = {k.split(".", 1)[1] for k in inject if "." in k and k.split(".", 1)[0] in tags}
    for sfx in suffixes:
        parts = [inject.get(f"{t}.{sfx}") for t in tags]
        if all(p is not None for p in parts):
            out[f"{prefix}.{sfx}"] = torch.cat(parts, dim=0).contiguous()
    return out


class GanTrainer:
    FUSED_D_HEAD = os.environ.get("TG_D_HEAD_FUSED", "1") != "0"      # the discriminator step's head + loss + head backward as one launch
    EARLY_SIDE_WORK = os.environ.get("TG_EARLY_SIDE_WORK", "1") != "0"   # gradient zeroing + discriminator dropout draws on the forward's audio fork

    def __init__(self, generator, discriminator, args, grad_sync=None):
        self.gen, self.dis = generator, discriminator
        self.G, self.D = generator.engine, discriminator.engine
        self.hp = hyper_params(args)
        # z_type variants (train_gan.py:59-84): 'speaker' = shuffled-speaker forward + div_reg + KLD, 'random' = second noise
        # forward + div_reg, anything else (or loss_reg_weight == 0) = regression (+ GAN) loss only
        self.z_type = getattr(args, "z_type", "speaker")
        assert (self.G.z_mode == "speaker") == (self.z_type == "speaker"), "args.z_type does not match the generator's z_obj"
        self.use_reg = self.z_type in ("speaker", "random") and self.hp["loss_reg_weight"] > 0.0 and self.G.z_mode is not None
        self.g_opt = FusedAdam(self.G, lr=self.hp["learning_rate"], betas=(0.5, 0.999))
        self.d_opt = FusedAdam(self.D, lr=self.hp["learning_rate"] * self.hp["discriminator_lr_weight"], betas=(0.5, 0.999))
        self.grad_sync = grad_sync          # ddp.GradSync or None
        if grad_sync is not None:
            # data parallel: the forward keeps the audio encoder on its second stream; in the backward the audio branch runs on the main stream
            # BEHIND the {text, speaker} bucket's hand-over, so that bucket's all-reduce (30 MB with the word embedding) has the audio
            # backward (~250 us) as its cover and a graph segment never ends with an un-joined branch.  TG_DDP_BWD_FORK=1: backward forked
            # too -- {audio} and {text, speaker} then become final together and leave in ONE exchange with nothing left to cover it
            # (GeneratorEngine.backward merges them); on one rank the two orders are level (4.90 / 4.91 ms against plain 4.63,
            # profiles/r5_g_ddp.txt: what the fork saves, the exposed 31 MB exchange costs), with real peers the covered order wins
            # Round 6: forked by default -- on one rank captured + forked is the fastest form (4.60 against 4.68 ms, profiles/r6_ddp.txt), and the
            # weight gradients' side rows (engine.GeneratorEngine.backward) ride on the same switch.  TG_DDP_BWD_FORK=0 restores the covered order.
            self.G.audio_fork_bwd = os.environ.get("TG_DDP_BWD_FORK", "1") != "0"
        self.keep_tape = False              # tests: keep the last stacked generator forward's tape in self.last_tape (holds its activations alive)
        self.last_tape = None
        self.prep = L.WeightPrep()          # transposed / packed weight operands, refreshed once per optimiser step
        self._cut = None                    # set by GraphedGanStep while capturing: cuts the graph at sync points

    def _sync(self, *action):
        """A gradient-exchange point.  Eager, or capturing with collectives inside the graph (GraphedGanStep, default): run the
        collective here -- RCCL's kernels become nodes of the same hipGraph, on RCCL's stream, joined back by the 'wait' action.
        Capturing in segment mode: end the current graph segment here and let the replay loop issue the collective between segments."""
        if self.grad_sync is None:
            return
        if self._cut is not None:
            self._cut(action)
        else:
            self.grad_sync.run(action)

    def _assert_no_pending_exchange(self):
        """The persistent cluster GRU kernels need all their workgroups co-resident (csrc/gru_cluster_x3.hip): no gradient bucket may
        still be in flight on RCCL's stream when one of them starts.  The schedule guarantees it (buckets are launched after the last
        recurrence of the backward, 'wait' precedes the optimiser step); this makes a future reordering fail loudly instead of
        stalling a cluster until its spin bound."""
        if self.grad_sync is not None and self.grad_sync.pending:
            raise RuntimeError("a gradient bucket is still in flight at a point where cluster-synchronised kernels are about to run")

    # ---- complete training state (weights, gradients, Adam moments and counters, RNG counters, BatchNorm buffers)
    def _state_tensors(self):
        out = []
        for eng in (self.G, self.D):
            s = eng.slab.ensure()
            out += [s.flat, s.grad, s.m, s.v, s.step, eng.rng.state]
            out += [b for _, b in eng.mod.named_buffers()]
        return out

    def snapshot(self):
        """Copies of every tensor an iteration reads AND writes; restore(snapshot()) makes the next iteration repeat the last one bit
        for bit up to the order of atomic float sums (used by the data-parallel self-check and the trajectory tests)."""
        return [t.detach().clone() for t in self._state_tensors()]

    def restore(self, snap):
        with torch.no_grad():
            for t, c in zip(self._state_tensors(), snap):
                t.copy_(c)

    # -------------------------------------------------------------------------------------------------------
    def train_iter(self, epoch, in_text, in_audio, target, vid, inject=None):
        """in_text (B,34) int64, in_audio (B,A) f32, target (B,34,27) f32, vid (B,) int64, all on the GPU.
        `inject` (tests only) replays recorded random draws; names follow oracle.ref_model.Rand."""
        hp = self.hp
        post = epoch > hp["loss_warmup"] and hp["loss_gan_weight"] > 0.0
        self.prep.add_slab("G", self.G.slab.ensure().flat)
        self.prep.add_slab("D", self.D.slab.ensure().flat)
        with self.prep.active():
            # both networks may have been changed since the last call (optimiser steps, load_state_dict): one batched launch each
            # (the operands that the main stream does not read before the forward's fork is joined are refreshed from phase_forward, on the
            # forked branch: layers.WeightPrep's parts)
            self.prep.late = False
            self.prep.refresh("G", "main0")
            st = self.phase_forward(post, in_text, in_audio, target, vid, inject)      # (refreshes the discriminator's operands too)
            self.prep.late = True
            if post:
                self.phase_d_step(st, inject)
                self.prep.refresh("D")                     # the discriminator's weights moved (train_gan.py:43)
            self.phase_g_backward(st, post, inject)
            self.phase_g_update()
        return StepLosses(st["g_scalars"], st.get("d_scalar"), hp, post)

    # ---- phase 1: stacked generator forward (train_gan.py:30,50,67)
    def phase_forward(self, post, in_text, in_audio, target, vid, inject):
        G, D = self.G, self.D
        B = target.shape[0]
        dev = target.device
        self._assert_no_pending_exchange()
        if G.use_side_stream and ops.GRU_CLUSTER:
            raise RuntimeError("use_side_stream cannot be combined with the cluster-synchronised GRU kernels (ops.GRU_CLUSTER): "
                               "side-stream kernels beside them break the co-residency their hand-off relies on")
        target = target.contiguous().float()
        speaker = self.G.z_mode == "speaker"
        tags = (["g1"] if post else []) + ["g2"] + (["g3"] if self.use_reg else [])
        ng, i2 = len(tags), tags.index("g2")
        # ONE launch: both RNG step counters and the Adam step counters of the optimisers that step in this iteration; the seed poses, word
        # ids and speaker ids of the ng stacked generator calls, the last call's ids shuffled by the diversity term's permutation (:67-72)
        perm_in = inject["perm"].to(dev).long().contiguous() if (inject is not None and "perm" in inject) else None
        # the discriminator step reads [target ; out1] (:30-31): one buffer, the head launch copies the target into its first B rows and the
        # generator's last layer writes its poses behind them, so the concatenation never runs
        d_in = torch.empty((1 + ng) * B, *target.shape[1:], device=dev) if post else None
        pre_s, text_s, vid_s = ops.iter_head(G.rng.state, D.rng.state, self.g_opt.slab.step, self.d_opt.slab.step if post else None, target,
                                             self.hp["n_pre_poses"], ng, text=in_text.contiguous(), vid=vid.contiguous() if speaker else None,
                                             permute_last=speaker and self.use_reg, perm_in=perm_in, perm_site=G.rng.site("perm"),
                                             row_floats=G.in_size,         # the seed poses land in the GRU input rows directly
                                             target_copy=d_in[:B] if post else None)
        # launches of the later phases that depend on nothing but the weights or the RNG state -- the refresh of the weight operands only the
        # backward passes and the discriminator read (layers.WeightPrep part "late"), zeroing both gradient slabs and the output MLP's
        # accumulators, the discriminator's dropout draws -- go out on the generator forward's forked audio branch (bandwidth-sized kernels beside
        # the text encoder's products); without that fork (text-only contexts, 8 hardware queues) the phases issue them themselves, where the
        # chain used to wait for them.  The operands the forward itself reads behind the join go first on that branch (side_head).
        early = {}
        def side_work():
            self.prep.refresh("G", "late")
            self.prep.refresh("D")                          # first read in the discriminator step / the generator step's D(out)
            G.slab.ensure().zero_grad()
            if ops.OUT_MLP_COMPOSED:
                early["out_acc"] = ops.zeros(G.pose_dim * G.H + G.pose_dim, device=dev)      # accumulators of the output MLP's backward
            if post:
                D.slab.ensure().zero_grad()
            if inject is None:
                early["d_out"] = D.draw_drop_masks(B, "d_out")
                if post:
                    early["d"] = D.draw_drop_masks(2 * B, "d")
        forks = self.EARLY_SIDE_WORK and hasattr(G, "forks_in_forward") and G.forks_in_forward(True)
        if not forks:
            self.prep.refresh("G", "side0")
        res = G.forward(pre_s, text_s, in_audio.float(), vid_s, training=True, groups=ng, save=True,
                        inject=_stack_inject(inject, tags, "g"), tag="g", save_rows=(i2 * B, B),     # only call g2 is differentiated (:50-88)
                        out_into=d_in[B:] if post else None, **(dict(side_work=side_work, side_head=lambda: self.prep.refresh("G", "side0")) if forks else {}))
        early["zeroed"] = res.get("side_work_ran", False)
        if not early["zeroed"]:
            self.prep.refresh("G", "late")
            self.prep.refresh("D")
        if self.keep_tape:
            self.last_tape = res["tape"]
        sl = lambda t, i: None if t is None else t[i * B:(i + 1) * B]
        st = dict(B=B, target=target, res=res, i2=i2, ng=ng, early=early, out2=sl(res["out"], i2), out3=sl(res["out"], ng - 1),
                  z2=sl(res["z"], i2), z3=sl(res["z"], ng - 1), mu2=sl(res["mu"], i2), lv2=sl(res["logvar"], i2))
        if post:
            st["d_in"] = d_in[:2 * B]                                   # [target ; out1]: D(real) first, then D(fake.detach())
        return st

    # ---- phase 2: discriminator step (train_gan.py:27-43)
    def phase_d_step(self, st, inject):
        D, B = self.D, st["B"]
        if not st["early"]["zeroed"]:
            D.slab.ensure().zero_grad()
        self._assert_no_pending_exchange()                    # the fused front-end kernels (csrc/d_preconv.hip) meet at device-wide barriers
        if self.FUSED_D_HEAD:
            # head forward + the clips' loss terms (:41) + head backward in one launch: dis_error is a mean of per-clip terms of the clip's own
            # logit; the mean itself is taken when the losses are read (StepLosses.to_dict)
            dres = D.forward(st["d_in"], training=True, groups=2, save=True, inject=_stack_inject(inject, ["d_real", "d_fake"], "d"), tag="d",
                             head_step=(B, 1.0 / B, 1.0 / B, True), drop_masks=st["early"].get("d"))
            st["d_scalar"] = dres["terms"]
            D.backward(dres["tape"], None, b0=0, nb=2 * B, param_grads=True)
        else:
            st["d_scalar"] = torch.empty(1, device=st["d_in"].device)
            dres = D.forward(st["d_in"], training=True, groups=2, save=True, inject=_stack_inject(inject, ["d_real", "d_fake"], "d"), tag="d",
                             drop_masks=st["early"].get("d"))
            logit = dres["logit"].view(-1)
            d_logit = torch.empty_like(logit)
            ops.gan_d_loss(logit[:B], logit[B:], st["d_scalar"], d_logit[:B], d_logit[B:])
            D.backward(dres["tape"], d_logit.view(-1, 1), b0=0, nb=2 * B, param_grads=True)
        self._sync("all", D.slab)
        self.d_opt.step(counter_advanced=True)

    # ---- phase 3: generator losses and backward (train_gan.py:47-91)
    def phase_g_backward(self, st, post, inject):
        G, D, B, hp = self.G, self.D, st["B"], self.hp
        dev = st["target"].device
        if not st["early"]["zeroed"]:
            G.slab.ensure().zero_grad()
        out2 = st["out2"].contiguous()
        self._assert_no_pending_exchange()
        # after warm-up the head's backward for the generator's GAN term (:57, 86-88) runs inside the head's forward launch (its d_logit
        # depends on the clip's own logit only); the loss kernel below still reads the logits for the gen_error scalar
        fused_head = post and self.FUSED_D_HEAD
        dres = D.forward(out2, training=True, groups=1, save=post, inject=inject, tag="d_out",   # runs in warm-up too (:55)
                         head_step=(B, hp["loss_gan_weight"] / B, 0.0, False) if fused_head else None, drop_masks=st["early"].get("d_out"))
        d_out = torch.empty_like(out2)
        d_logit = torch.empty(B, device=dev)
        st["g_scalars"] = torch.empty(5, device=dev)
        # the fused loss kernel covers every z_type through its weights: terms the reference leaves out (:59-84) get weight 0
        # and neutral operands (out_rand = out, z_rand = z, mu = logvar = 0)
        speaker_terms = self.G.z_mode == "speaker" and self.use_reg
        zero_z = ops.zeros(B, 16, device=dev) if not (speaker_terms and self.use_reg) else None      # neutral operand of the terms left out
        mu2, lv2 = (st["mu2"].contiguous(), st["lv2"].contiguous()) if speaker_terms else (zero_z, zero_z)
        z2, z3 = (st["z2"].contiguous(), st["z3"].contiguous()) if self.use_reg else (zero_z, zero_z)
        out3 = st["out3"].contiguous() if self.use_reg else out2
        d_mu, d_lv = torch.empty(B, 16, device=dev), torch.empty(B, 16, device=dev)
        ops.gan_g_loss(out2, st["target"], out3, z2, z3, mu2, lv2, dres["logit"].view(-1),
                       (hp["loss_regression_weight"], hp["loss_kld_weight"] if speaker_terms else 0.0,
                        hp["loss_reg_weight"] if self.use_reg else 0.0, hp["loss_gan_weight"]),
                       post, torch.empty(3 * B, device=dev), st["g_scalars"], d_out, d_mu, d_lv, d_logit)
        if not speaker_terms:
            d_mu = d_lv = None
        if post:
            D.backward(dres["tape"], None if fused_head else d_logit.view(B, 1), param_grads=False, need_dposes=True,
                       dposes_into=d_out)                                                                 # d_out += dD/dposes (:86-88)
        self._assert_no_pending_exchange()                    # the generator's backward recurrences come next
        # the audio encoder's bucket is the backward's last (engine.GeneratorEngine.backward): launched and awaited in ONE action, so that a
        # segmented graph is cut once there, not twice around an empty segment
        self._waited = False
        def on_ready(prefixes, last=False):
            # `last` comes from the engine (the bucket it hands over at the very end of its backward), not from the bucket's name
            assert not self._waited, "a gradient bucket was handed over after the backward's last one"
            self._sync("bucket_wait" if last else "bucket", G.slab, prefixes)
            self._waited = last
        G.backward(st["res"]["tape"], d_out, d_mu, d_lv, b0=st["i2"] * B, nb=B, on_ready=on_ready if self.grad_sync is not None else None,
                   **({"out_acc": st["early"]["out_acc"]} if "out_acc" in st["early"] else {}))

    # ---- phase 4: generator update (train_gan.py:92)
    def phase_g_update(self):
        if not getattr(self, "_waited", False):
            self._sync("wait")
        self._waited = False
        if self.grad_sync is not None and self._cut is None:
            assert not self.grad_sync.pending, "a gradient bucket is still in flight at the optimiser step"
        self.g_opt.step(counter_advanced=True)


class GraphedGanStep:
    """Captures GanTrainer.train_iter for fixed shapes into a hipGraph and replays it.

    Inputs are copied into static buffers; every intermediate lives in the graphs' private memory pool; random draws
    advance through the device-side Philox step counter, Adam through its device-side step counter, so every replay is a
    new, correct training iteration.  Under data parallelism the iteration is captured as several graph SEGMENTS cut at the
    gradient-exchange points: the replay loop launches each segment and issues the RCCL all-reduce of the bucket that just
    became final, asynchronously, so it overlaps the next segment (the rest of the backward)."""

    def __init__(self, trainer: GanTrainer, epoch, in_text, in_audio, target, vid, warmup_iters=2, capture_collectives=None):
        """capture_collectives (data parallel only): True = the RCCL all-reduces are captured INTO the graph (one graph per iteration,
        no host work between segments); False = graph segments cut at the exchange points, collectives issued eagerly between them.
        Default: environment TG_DDP_CAPTURE (0 unless set to 1).  Segments are the default since round 3: with collectives recorded inside a
        capture, RCCL's watchdog thread was seen to abort the process ("operation not permitted on an event last recorded in a capturing
        stream" from WorkNCCL::isCompleted -- an event of torch's cache that a captured collective had recorded, queried later through an
        eager one); it cannot be caught in Python, and costs more than the 1.3 % the single graph saves."""
        import os
        if capture_collectives is None:
            capture_collectives = os.environ.get("TG_DDP_CAPTURE", "0") != "0"
        self.capture_collectives = bool(capture_collectives) and trainer.grad_sync is not None
        self.trainer, self.epoch = trainer, epoch
        # the step's inputs live in ONE buffer (views): a feeder moves a whole batch in with a single copy (data.DeviceBatchFeeder)
        from .data import packed_like
        self.static_flat, views = packed_like((in_text, in_audio, target, vid))
        self.static = list(views)
        for dst, src in zip(self.static, (in_text, in_audio, target, vid)):
            dst.copy_(src)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                                   # warm-up outside capture (allocator, lazy init)
            for _ in range(warmup_iters):
                trainer.train_iter(epoch, *self.static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.segments = []                   # [(graph, action to run after it or None)]
        pool = torch.cuda.graph_pool_handle()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            state = {"g": torch.cuda.CUDAGraph()}
            # thread-local capture mode: other threads (the RCCL watchdog polls its events with hipEventQuery) must not invalidate
            # the capture; everything this thread issues between begin and end is still checked
            # (also when a process group merely EXISTS in this process: RCCL's helper threads -- communicator init, watchdog -- call the HIP
            # runtime on their own schedule, and under the global mode such a call during our capture invalidates it or aborts the process:
            # seen once as a SIGABRT while capturing a trainer without grad_sync after init_process_group("nccl"))
            pg_alive = torch.distributed.is_available() and torch.distributed.is_initialized()
            mode = "thread_local" if (trainer.grad_sync is not None or pg_alive) else "global"
            state["g"].capture_begin(pool=pool, capture_error_mode=mode)

            def cut(action):
                state["g"].capture_end()
                self.segments.append((state["g"], action))
                state["g"] = torch.cuda.CUDAGraph()
                state["g"].capture_begin(pool=pool, capture_error_mode=mode)
            trainer._cut = None if self.capture_collectives else cut
            try:
                self.losses = trainer.train_iter(epoch, *self.static)
            finally:
                trainer._cut = None
                state["g"].capture_end()
            self.segments.append((state["g"], None))
        torch.cuda.current_stream().wait_stream(cap)

    CHECK_EVERY = 16      # replays between two reads of the persistent kernels' sticky timeout word (one host sync each)

    def __call__(self, in_text=None, in_audio=None, target=None, vid=None):
        self._n_replays = getattr(self, "_n_replays", 0) + 1
        if self._n_replays % self.CHECK_EVERY == 0:
            ops.check_async_errors()       # a timed-out recurrence must not train on garbage until somebody reads the losses
        for dst, src in zip(self.static, (in_text, in_audio, target, vid)):
            if src is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        for graph, action in self.segments:
            graph.replay()
            if action is not None:
                self.trainer.grad_sync.run(action)
        return self.losses


def _all_ranks_agree(ok, device):
    """Logical AND of `ok` over the process group (every rank must take the same path: a rank replaying captured collectives beside one
    issuing them eagerly deadlocks)."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return bool(ok)
    backend = torch.distributed.get_backend()
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
    return bool(int(flag.item()))


def checked_ddp_step(trainer: GanTrainer, epoch, in_text, in_audio, target, vid, warmup_iters=2, tol=1e-3, modes=None, log=None):
    """The data-parallel graphed step, self-checked before anybody times or trains with it (scripts/train.py:93-96 is the reference's
    multi-GPU switch; its DataParallel has no such failure mode -- captured RCCL collectives do).

    One EAGER iteration from the current state gives the reference losses; then, per candidate mode in order -- collectives captured
    inside the hipGraph (default first), graph segments with eager collectives between them -- the graph is built, the state restored,
    ONE replay run and its losses compared with the eager ones (relative `tol`).  A mode is accepted only if capture raised on NO rank
    and the losses agree on EVERY rank (agreement is all-reduced, so all ranks switch together, in-process).  The state is restored
    afterwards: the caller starts from exactly the state it passed in.  Returns (step, info) with info = {"ranks", "collectives",
    "rejected": [(mode, reason)]}."""
    import os
    assert trainer.grad_sync is not None, "checked_ddp_step is the data-parallel constructor"
    dev = target.device
    say = log or (lambda *_: None)
    if modes is None:
        modes = ["captured", "segments"] if os.environ.get("TG_DDP_CAPTURE", "0") != "0" else ["segments"]      # (see GraphedGanStep)
    snap = trainer.snapshot()
    ref = trainer.train_iter(epoch, in_text, in_audio, target, vid).to_dict()
    trainer.restore(snap)
    rejected = []
    for mode in modes:
        step, why = None, None
        try:
            step = GraphedGanStep(trainer, epoch, in_text, in_audio, target, vid, warmup_iters=warmup_iters,
                                  capture_collectives=(mode == "captured"))
        except Exception as e:                       # capture refused (RCCL / runtime): fall through to the next mode
            why = f"capture raised {type(e).__name__}: {e}"
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
        if not _all_ranks_agree(step is not None, dev):
            rejected.append((mode, why or "capture failed on another rank"))
            say(f"ddp self-check: {mode} rejected ({rejected[-1][1]})")
            trainer.grad_sync.pending = []
            trainer.restore(snap)
            continue
        trainer.restore(snap)
        got = step().to_dict()
        bad = [k for k in ref if k not in got or not (abs(got[k] - ref[k]) <= tol * max(1.0, abs(ref[k])))]
        if not _all_ranks_agree(not bad, dev):
            why = "losses differ from the eager iteration: " + ", ".join(f"{k} {got.get(k)} vs {ref[k]}" for k in bad) if bad else \
                  "losses differ on another rank"
            rejected.append((mode, why))
            say(f"ddp self-check: {mode} rejected ({why})")
            trainer.restore(snap)
            continue
        trainer.restore(snap)
        say(f"ddp self-check: {mode} accepted, losses equal the eager iteration within {tol:g}")
        return step, {"ranks": trainer.grad_sync.world, "collectives": mode, "rejected": rejected}
    raise RuntimeError("no data-parallel graph mode passed its self-check: " + "; ".join(f"{m}: {w}" for m, w in rejected))
