"""Training loop with the torchgan ``Trainer`` interface the reference drives
(src/histopathology_gan.py:298-314, src/gan_utils.py:286-297).  torchgan itself is third-party
and absent from the reference tree; behaviour follows SURVEY.md Appendix A:

  * ``Trainer(models, losses_list, ..., device, ncritic, epochs, sample_size, checkpoints,
    retain_checkpoints, recon, test_noise, nrow, **kwargs)``; unknown kwargs become attributes
    (the reference passes ``devices=[0]``);
  * models / optimizers are built from ``{"name": cls, "args": {...}, "optimizer": {"name": cls,
    "args": {...}}}`` and exposed as attributes ``generator``, ``optimizer_generator``, ...;
  * per batch every loss's ``train_ops`` is called with arguments resolved BY NAME from the
    trainer's attributes (``loss.arg_map`` may rename); losses run in list order, generator losses
    only every ``ncritic`` discriminator iterations;
  * per epoch: checkpoint ``<checkpoints><k>.model`` (k cycling), console summary, sample grid
    ``<recon>/epoch<e+1>_generator.png`` from the generator in eval mode;
  * checkpoint dict keys: epoch, loss_information, loss_objects, metric_objects, loss_logs,
    metric_logs, <model>, optimizer_<model>.

Data parallel: one process per GPU; rank 0's parameters are broadcast at start, gradients are
all-reduced inside the losses, only rank 0 writes checkpoints / images.
"""
from __future__ import annotations

import os
from inspect import signature

import torch

from . import dist as D_
from . import losses as L
from . import optim
from .prefetch import DevicePrefetcher


class Trainer:
    def __init__(self, models, losses_list, metrics_list=None, device=torch.device("cuda:0"), ncritic=1, epochs=5,
                 sample_size=8, checkpoints="./model/gan", retain_checkpoints=5, recon="./images", log_dir=None,
                 test_noise=None, nrow=8, precision="bf16", prefetch=True, **kwargs):
        self.device = torch.device(device)
        self.prefetch = bool(prefetch)          # extra knob: host batches are copied to the device one iteration ahead
        self.pipeline = bool(kwargs.pop("pipeline", True))   # extra knob: see train_iter
        self.model_names = []
        self.optimizer_names = []
        self.schedulers = []
        for key, cfg in models.items():
            self.model_names.append(key)
            model = cfg["name"](**cfg.get("args", {}))
            if hasattr(model, "set_precision"):
                model.set_precision(precision)
            model = model.to(self.device)
            setattr(self, key, model)
            opt_cfg = cfg["optimizer"]
            opt_cls = opt_cfg["name"]
            if opt_cls is torch.optim.Adam:
                opt_cls = optim.Adam                       # same update rule, fused HIP kernel
            opt = opt_cls(model.parameters(), **opt_cfg.get("args", {}))
            if isinstance(opt, optim.Adam):
                opt.bind(model)
            opt_name = "optimizer_{}".format(key)
            setattr(self, opt_name, opt)
            self.optimizer_names.append(opt_name)
            if "scheduler" in cfg:
                sch = cfg["scheduler"]
                self.schedulers.append(sch["name"](opt, **sch.get("args", {})))
        for m in self.model_names:                         # identical replicas in a data-parallel run
            mod = getattr(self, m)
            for t in list(mod.parameters()) + list(mod.buffers()):
                D_.broadcast_(t.data, 0)
        self.losses = {}
        for loss in losses_list:
            self.losses[type(loss).__name__] = loss
            bv = getattr(loss, "betavae", None)
            if precision == "fp16" and bv is not None and hasattr(bv, "set_precision"):
                bv.set_precision("fp16")                  # the frozen encoder's GEMMs in the same 16-bit type as the GAN (configs[3])
        self.metrics = {} if metrics_list is None else {type(m).__name__: m for m in metrics_list}
        self.sample_size = sample_size
        self.nrow = nrow
        self.checkpoints = checkpoints
        self.retain_checkpoints = retain_checkpoints
        self.recon = recon
        self.log_dir = log_dir
        self.test_noise = test_noise
        self.loss_information = {"generator_losses": 0.0, "discriminator_losses": 0.0,
                                 "generator_iters": 0, "discriminator_iters": 0}
        self.loss_logs = {name: [] for name in self.losses}
        self.metric_logs = {}
        self.ncritic = ncritic
        self.start_epoch = 0
        self.last_retained_checkpoint = 0
        self.epochs = epochs
        self.batch_size = None
        self.real_inputs = None
        self.labels = None
        self.noise = None
        for k, v in kwargs.items():
            if k not in self.__dict__:
                setattr(self, k, v)
        os.makedirs(os.path.dirname(self.checkpoints) or ".", exist_ok=True) if D_.rank() == 0 else None
        if D_.rank() == 0 and self.recon:
            os.makedirs(self.recon, exist_ok=True)

    # ------------------------------------------------------------------ checkpoints
    def save_model(self, epoch, save_items=None):
        from .ops_hip import check_handoffs
        check_handoffs()            # never checkpoint weights behind a fused-kernel hand-off that timed out (every rank)
        if D_.rank() != 0:
            return
        if self.last_retained_checkpoint == self.retain_checkpoints:
            self.last_retained_checkpoint = 0
        save_path = self.checkpoints + str(self.last_retained_checkpoint) + ".model"
        self.last_retained_checkpoint += 1
        print("Saving Model at '{}'".format(save_path))
        model = {"epoch": epoch + 1, "loss_information": self.loss_information, "loss_objects": self.losses,
                 "metric_objects": self.metrics, "loss_logs": self.loss_logs, "metric_logs": self.metric_logs}
        for save_item in self.model_names + self.optimizer_names:
            model.update({save_item: getattr(self, save_item).state_dict()})
        if save_items is not None:
            for it in ([save_items] if isinstance(save_items, str) else save_items):
                model.update({it: getattr(self, it)})
        torch.save(model, save_path)

    def load_model(self, load_path="", load_items=None):
        if load_path == "":
            load_path = self.checkpoints + str(self.last_retained_checkpoint) + ".model"
        print("Loading Model From '{}'".format(load_path))
        try:
            # A checkpoint written by the reference pickles its live plugin objects (torchgan.losses.*, wgan_loss.*LossVAE
            # holding a betaVAE.betaVAE each) under loss_objects / metric_objects: a plain torch.load raises
            # ModuleNotFoundError on the first such class, before the dictionary exists.  The tolerant unpickler resolves
            # every unimportable global to an inert placeholder; the live plugin objects of THIS trainer are kept
            # (SURVEY 5 / 8b "Checkpoint": tolerate missing / unknown loss_objects, metric_*).
            from . import _tolerant_pickle
            before = set(_tolerant_pickle.missing_globals())
            checkpoint = torch.load(load_path, map_location="cpu", weights_only=False, pickle_module=_tolerant_pickle)
            absent = [k for k in _tolerant_pickle.missing_globals() if k not in before]
            if absent:      # say what was replaced: the reference's plugin classes are expected, anything else is worth a look
                print("load_model: {} global(s) of the checkpoint are not importable here and were read as inert "
                      "placeholders: {}".format(len(absent), ", ".join("%s.%s" % k for k in absent[:12]) +
                                                (" ..." if len(absent) > 12 else "")))
            self.start_epoch = checkpoint["epoch"]
            self.loss_information = checkpoint.get("loss_information", self.loss_information)
            self.loss_logs = checkpoint.get("loss_logs", self.loss_logs)
            for name in self.losses:                       # a log keyed by plugin names this trainer does not know stays
                self.loss_logs.setdefault(name, [])        # (torchgan keeps it too); every live plugin needs its list
            self.metric_logs = checkpoint.get("metric_logs", self.metric_logs)
            for load_item in self.model_names + self.optimizer_names:
                getattr(self, load_item).load_state_dict(checkpoint[load_item])
            if load_items is not None:
                for it in ([load_items] if isinstance(load_items, str) else load_items):
                    obj = checkpoint[it]
                    inner = list(obj.values()) if isinstance(obj, dict) else list(obj) if isinstance(obj, (list, tuple)) else []
                    for o in [obj] + inner:               # the item itself or a member of a plain container
                        if _tolerant_pickle.is_placeholder(o):
                            cls = o if isinstance(o, type) else type(o)
                            raise RuntimeError("load_items: '%s' holds an object pickled as %s.%s, which is not importable "
                                               "here" % ((it,) + tuple(cls._rg_missing)))
                    setattr(self, it, obj)
        except Exception as e:  # torchgan prints and continues; a silent restart would hide corruption
            raise RuntimeError("Model could not be loaded from {}: {}".format(load_path, e))

    # ------------------------------------------------------------------ loop
    def _get_argument_maps(self, default_map, func):
        sig = signature(func)
        arg_map = dict(default_map)
        for sig_param in sig.parameters:
            if sig_param not in arg_map:
                arg_map[sig_param] = sig_param
        return arg_map

    def _get_arguments(self, arg_map):
        return dict(zip(arg_map.keys(), map(lambda x: getattr(self, x), arg_map.values())))

    def _store_loss_maps(self):
        self._arg_maps = {name: self._get_argument_maps(loss.arg_map, loss.train_ops)
                          for name, loss in self.losses.items()}
        # a gradient-penalty plugin directly behind the D-loss plugin (the reference's loss list): the D-loss plugin then
        # draws the penalty's noise right after its own and produces both fake batches in one generator pass
        names = list(self.losses.values())
        L.TRAINER_LOOKAHEAD[0] = any(
            isinstance(a, (L.WassersteinDiscriminatorLoss, L.WassersteinDiscriminatorLossVAE)) and
            isinstance(b, (L.WassersteinGradientPenalty, L.WassersteinGradientPenaltyVAE)) and
            isinstance(a, L._VAEMixin) == isinstance(b, L._VAEMixin)
            for a, b in zip(names, names[1:]))

    def train_iter(self, carry=False):
        """One batch through every loss plugin, in list order.  The library's own plugins split their
        ``train_ops`` into an asynchronous launch (``train_ops_async``: all host-side work -- argument checks, random
        draws, graph launch -- returning the loss as a device scalar) and the ``.item()`` that ``train_ops`` ends with.
        With ``pipeline`` (default) the value of train_op k (copied to a pinned host slot right behind it, ``_post``) is read AFTER train_op k + 1 has been launched: same
        launches, same draws, same values in the same order in ``loss_logs``, but the device queue never runs dry while the
        host prepares the next launch (three idle gaps of 0.2-0.3 ms per iteration otherwise).  A plugin whose
        ``train_ops`` is not the library's (a subclass override, a user plugin) is called as is.
        carry (used by train()): the LAST train_op's .item() is left pending and taken by the next call after its first
        launch (``flush_pending()`` takes it at the end of an epoch); its value is then part of the next call's sums -- the
        epoch totals and ``loss_logs`` are unchanged."""
        acc = {"g": 0.0, "d": 0.0}
        gen_iter, dis_iter = 0, 0
        pending = self.__dict__.setdefault("_pending", [])

        def finish():
            while pending:
                name, kind, val = pending.pop(0)
                cur = self._take(val)
                self.loss_logs[name].append(cur)
                acc[kind] += cur

        L.new_batch()          # the conditioning latent is encoded once per batch and shared by the loss plugins
        for name, loss in self.losses.items():
            if isinstance(loss, L.GeneratorLoss) and isinstance(loss, L.DiscriminatorLoss):
                raise NotImplementedError("joint generator/discriminator losses are not on the RNA-GAN path")
            if isinstance(loss, L.GeneratorLoss):
                if self.loss_information["discriminator_iters"] % self.ncritic != 0:
                    continue
                kind = "g"
                gen_iter += 1
            elif isinstance(loss, L.DiscriminatorLoss):
                kind = "d"
                dis_iter += 1
            else:
                continue
            twin = getattr(type(loss).train_ops, "_rg_async", None) if getattr(self, "pipeline", True) else None
            fn = getattr(loss, twin) if twin else loss.train_ops
            val = fn(**self._get_arguments(self._arg_maps[name]))
            if twin:
                val = self._post(val)
            finish()           # the PREVIOUS train_op's value: its copy to the host was enqueued before the launch just made
            pending.append((name, kind, val))
        if not carry:
            finish()
        return acc["g"], acc["d"], gen_iter, dis_iter

    def _post(self, val):
        """Enqueue the copy of a train_op's device scalar into a pinned host slot right behind that train_op (a later
        ``tensor.item()`` would be ordered behind whatever has been launched since -- i.e. behind the NEXT train_op); returns
        (slot, event)."""
        if not (torch.is_tensor(val) and val.is_cuda):
            return val
        ring = self.__dict__.setdefault("_host_ring", [])
        if not ring:
            for _ in range(4):
                ring.append((torch.empty(1, dtype=torch.float32, pin_memory=True), torch.cuda.Event()))
            self._host_ring_pos = 0
        slot, ev = ring[self._host_ring_pos % len(ring)]
        self._host_ring_pos += 1
        slot.copy_(val.detach().reshape(1).float(), non_blocking=True)
        ev.record(torch.cuda.current_stream(val.device))
        return slot, ev

    @staticmethod
    def _take(val):
        if isinstance(val, tuple):
            slot, ev = val
            ev.synchronize()
            return slot.item()
        return val.item() if torch.is_tensor(val) else val

    def flush_pending(self):
        """Take the value a train_iter(carry=True) left pending; returns its (generator, discriminator) loss sums."""
        acc = {"g": 0.0, "d": 0.0}
        pending = self.__dict__.setdefault("_pending", [])
        while pending:
            name, kind, val = pending.pop(0)
            cur = self._take(val)
            self.loss_logs[name].append(cur)
            acc[kind] += cur
        return acc["g"], acc["d"]

    def _poll_handoffs(self, final=False):
        """Every ``handoff_check_every`` iterations (default 64; 0 = only at save_model): start a sync-free read of the fused
        split-K BatchNorm kernels' error word (ops_hip.handoffs_poll_start: MAX over the ranks under data parallelism, copied
        to a pinned slot behind this iteration's launches) and look at the one started earlier -- a timed-out in-kernel
        rendezvous stops the run within that many iterations, on every rank together, instead of poisoning an epoch."""
        from . import ops_hip as H
        every = int(getattr(self, "handoff_check_every", 64))
        st = self.__dict__.setdefault("_handoff_state", {"it": 0, "pending": None})
        st["it"] += 1
        if st["pending"] is not None and (final or st["it"] % max(every, 1) == 1 or every == 1):
            h, st["pending"] = st["pending"], None
            H.handoffs_poll_finish(h)
        if not final and every > 0 and st["it"] % every == 0:
            st["pending"] = H.handoffs_poll_start()

    def sample_images(self, epoch):
        if D_.rank() != 0 or not self.recon:
            return
        gen = self.generator
        was_training = gen.training
        gen.eval()
        with torch.no_grad():
            img = gen(self.test_noise[0] if isinstance(self.test_noise, (list, tuple)) else self.test_noise)
        gen.train(was_training)
        save_image_grid(img, "{}/epoch{}_generator.png".format(self.recon, epoch + 1), nrow=self.nrow)

    def train(self, data_loader, **kwargs):
        for name in self.model_names:
            getattr(self, name).train()
        self._store_loss_maps()
        if self.test_noise is None:
            self.test_noise = self.generator.sampler(self.sample_size, self.device)
        for epoch in range(self.start_epoch, self.epochs):
            for name in self.model_names:
                getattr(self, name).train()
            # batches arrive on the device one iteration ahead (rna_gan_amd/prefetch.py); the .to() calls below and the
            # plugins' own then find the tensors in place
            for data in (DevicePrefetcher(data_loader, self.device) if self.prefetch else data_loader):
                if isinstance(data, (tuple, list)):
                    self.real_inputs = data[0].to(self.device)
                    self.labels = data[1].to(self.device)
                elif isinstance(data, torch.Tensor):
                    self.real_inputs = data.to(self.device)
                else:
                    self.real_inputs = data
                lgen, ldis, gen_iter, dis_iter = self.train_iter(carry=getattr(self, "pipeline", True))
                self._poll_handoffs()
                self.loss_information["generator_losses"] += lgen
                self.loss_information["discriminator_losses"] += ldis
                self.loss_information["generator_iters"] += gen_iter
                self.loss_information["discriminator_iters"] += dis_iter
            lgen, ldis = self.flush_pending()
            self._poll_handoffs(final=True)
            self.loss_information["generator_losses"] += lgen
            self.loss_information["discriminator_losses"] += ldis
            self.save_model(epoch)
            if D_.rank() == 0:
                gi = max(self.loss_information["generator_iters"], 1)
                di = max(self.loss_information["discriminator_iters"], 1)
                print("Epoch {} Summary\ngenerator Mean Loss : {}\ndiscriminator Mean Loss : {}".format(
                    epoch + 1, self.loss_information["generator_losses"] / gi,
                    self.loss_information["discriminator_losses"] / di))
            self.sample_images(epoch)
            for sch in self.schedulers:
                sch.step()
        print("Training of the Model is Complete") if D_.rank() == 0 else None

    def __call__(self, data_loader, **kwargs):
        self.batch_size = data_loader.batch_size
        self.train(data_loader, **kwargs)


def save_image_grid(img, path, nrow=8, pad=2):
    """torchvision.utils.save_image(img, path, nrow, normalize=True) without torchvision: min-max
    normalisation over the whole batch, zero-padded grid, 8-bit PNG."""
    from PIL import Image
    x = img.detach().float().cpu()
    lo, hi = float(x.min()), float(x.max())
    x = (x - lo) / max(hi - lo, 1e-5)
    n, c, h, w = x.shape
    ncol = min(nrow, n)
    nrows = (n + ncol - 1) // ncol
    grid = torch.zeros(c, nrows * (h + pad) + pad, ncol * (w + pad) + pad)
    for i in range(n):
        r, q = divmod(i, ncol)
        grid[:, pad + r * (h + pad): pad + r * (h + pad) + h, pad + q * (w + pad): pad + q * (w + pad) + w] = x[i]
    arr = (grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8)).numpy()
    if c == 1:
        arr = arr[:, :, 0]
    Image.fromarray(arr).save(path)
