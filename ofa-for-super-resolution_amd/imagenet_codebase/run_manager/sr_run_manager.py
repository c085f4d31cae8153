"""RunConfig + SRRunManager: the trainer API of the SR path (mirror of reference
ofa/imagenet_codebase/run_manager/sr_run_manager.py:25-549), re-hosted for one process per GPU.

Differences that are deliberate (MI355X-first) and do not change results on one GPU:
  * multi-GPU is process-per-GPU data parallelism with ONE flat RCCL all-reduce per optimizer step
    (distributed.FlatGradReducer) instead of single-process nn.DataParallel (:197-198); `num_gpus` is
    accepted for signature compatibility, the world size comes from torch.distributed;
  * the PSNR logged during training is computed on the device with the reference's exact formula
    (utils.psnr_y_device) so there is no per-sub-step host sync (:496 -> .cpu() every step);
  * apex / tensorboardX hooks are not carried over (both absent; dead code in the reference).
Checkpoint files, dict keys, log files and the optimizer grouping are the reference's.
"""
import json
import math
import os
import time

import numpy as np
import torch
import torch.nn as nn

from ... import distributed as dd
from ... import ops
from ...graphed import GraphedEval
from ...utils import AverageMeter, bucket_by_size, device_batch, psnr_y, psnr_y_per_image
from ..utils import get_net_info


class RunConfig(object):
    """reference :25-133"""

    def __init__(self, n_epochs, init_lr, lr_schedule_type, lr_schedule_param, dataset, train_batch_size,
                 test_batch_size, valid_size, opt_type, opt_param, weight_decay, label_smoothing, no_decay_keys,
                 mixup_alpha, model_init, validation_frequency, print_frequency):
        self.n_epochs = n_epochs
        self.init_lr = init_lr
        self.lr_schedule_type = lr_schedule_type
        self.lr_schedule_param = lr_schedule_param
        self.dataset = dataset
        self.train_batch_size = train_batch_size
        self.test_batch_size = test_batch_size
        self.valid_size = valid_size
        self.opt_type = opt_type
        self.opt_param = opt_param
        self.weight_decay = weight_decay
        self.label_smoothing = label_smoothing
        self.no_decay_keys = no_decay_keys
        self.mixup_alpha = mixup_alpha
        self.model_init = model_init
        self.validation_frequency = validation_frequency
        self.print_frequency = print_frequency

    @property
    def config(self):
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_")}

    def copy(self):
        return RunConfig(**self.config)

    # -- learning rate (cosine over all iterations, linear warm-up) -- reference :62-90
    def calc_learning_rate(self, epoch, batch=0, nBatch=None):
        if self.lr_schedule_type == "cosine":
            total, cur = self.n_epochs * nBatch, epoch * nBatch + batch
            return 0.5 * self.init_lr * (1 + math.cos(math.pi * cur / total))
        if self.lr_schedule_type is None:
            return self.init_lr
        raise ValueError("do not support: %s" % self.lr_schedule_type)

    def adjust_learning_rate(self, optimizer, epoch, batch=0, nBatch=None):
        new_lr = self.calc_learning_rate(epoch, batch, nBatch)
        for group in optimizer.param_groups:
            group["lr"] = new_lr
        return new_lr

    def warmup_adjust_learning_rate(self, optimizer, T_total, nBatch, epoch, batch=0, warmup_lr=0):
        cur = epoch * nBatch + batch + 1
        new_lr = cur / T_total * (self.init_lr - warmup_lr) + warmup_lr
        for group in optimizer.param_groups:
            group["lr"] = new_lr
        return new_lr

    # -- data provider
    @property
    def data_provider(self):
        raise NotImplementedError

    @property
    def train_loader(self):
        return self.data_provider.train

    @property
    def valid_loader(self):
        return self.data_provider.valid

    @property
    def test_loader(self):
        return self.data_provider.test

    def random_sub_train_loader(self, n_images, batch_size, num_worker=None, num_replicas=None, rank=None):
        return self.data_provider.build_sub_train_loader(n_images, batch_size, num_worker, num_replicas, rank)

    # -- optimizer (reference :115-133; Adam ignores momentum / nesterov there too)
    def build_optimizer(self, net_params):
        if self.no_decay_keys is not None:
            assert isinstance(net_params, list) and len(net_params) == 2
            groups = [{"params": net_params[0], "weight_decay": self.weight_decay},
                      {"params": net_params[1], "weight_decay": 0}]
        else:
            groups = [{"params": net_params, "weight_decay": self.weight_decay}]
        if self.opt_type == "sgd":
            opt_param = self.opt_param or {}
            return torch.optim.SGD(groups, self.init_lr, momentum=opt_param.get("momentum", 0.9),
                                   nesterov=opt_param.get("nesterov", True))
        if self.opt_type == "adam":
            # same update rule as the reference's torch.optim.Adam(groups, lr); on the GPU torch's fused multi-tensor
            # implementation does it in 2 launches instead of ~20 (parameters without a gradient are skipped either way)
            groups = [dict(g, params=list(g["params"])) for g in groups]   # the reference passes generators
            plist = [p for g in groups for p in g["params"]]
            fused = bool(plist) and all(p.is_cuda for p in plist) and os.environ.get("OFASR_FUSED_ADAM", "1") != "0"
            return torch.optim.Adam(groups, self.init_lr, fused=True if fused else None)
        raise NotImplementedError


class SRRunManager(object):
    """reference :136-549.  attrs: net, network, optimizer, run_config, device, best_acc, start_epoch, path,
    train_criterion, test_criterion; plus `reducer` (None on one GPU)."""

    def __init__(self, path, net, run_config, init=True, measure_latency=None, no_gpu=False, mix_prec=None,
                 num_gpus=None, args=None):
        self.path = path
        self.net = net
        self.run_config = run_config
        self.mix_prec = mix_prec   # None | 'bf16' | 'f16': activation dtype via torch.autocast (apex in the reference)
        self.best_acc = 0
        self.start_epoch = 0
        os.makedirs(self.path, exist_ok=True)
        ops.single_thread_backward(True)   # one process drives one GPU: backward() on the calling thread (ops.py)

        if no_gpu or not torch.cuda.is_available():
            raise RuntimeError("SRRunManager drives the MI355X HIP hot path: a GPU is required (no CPU fallback)")
        self.device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(self.device)
        self.net = self.net.to(self.device)
        torch.backends.cudnn.benchmark = True

        if init:
            self.network.init_model(run_config.model_init)
        dd.broadcast_module(self.network)

        if self.is_root:
            net_info = get_net_info(self.net, self.run_config.data_provider.data_shape, measure_latency, True, args)
            with open("%s/net_info.txt" % self.path, "w") as fout:
                fout.write(json.dumps(net_info, indent=4) + "\n")
                try:
                    fout.write(self.network.module_str)
                except Exception:
                    pass

        self.train_criterion = nn.MSELoss()
        self.test_criterion = nn.MSELoss()

        if self.run_config.no_decay_keys:
            keys = self.run_config.no_decay_keys.split("#")
            net_params = [list(self.network.get_parameters(keys, mode="exclude")),   # with weight decay
                          list(self.network.get_parameters(keys, mode="include"))]   # without
        else:
            net_params = list(self.network.weight_parameters())
        self.optimizer = self.run_config.build_optimizer(net_params)
        # OFASR_DP_OVERLAP=1: two buckets, the decoder tail's exchanged while the MB stack's backward still runs
        # (distributed.FlatGradReducer(early_params=...)); default: the single flat bucket
        early = None
        if os.environ.get("OFASR_DP_OVERLAP", "0") != "0" and hasattr(self.network, "early_gradient_parameters"):
            early = self.network.early_gradient_parameters()
        self.reducer = dd.FlatGradReducer(self.network.parameters(), gather=True, early_params=early) \
            if dd.is_distributed() else None

    # ------------------------------------------------------------------ distributed helpers
    @property
    def is_root(self):
        return dd.rank() == 0

    def zero_grad(self):
        """optimizer.zero_grad() on one GPU; re-arm the flat gradient bucket under data parallelism."""
        if self.reducer is not None:
            self.reducer.prepare()
        else:
            self.optimizer.zero_grad(set_to_none=True)

    def last_backward_next(self):
        """the next backward pass is the last one before step(): the overlapped gradient exchange may start in it"""
        if self.reducer is not None:
            self.reducer.arm()

    def step(self):
        """gradient exchange (one flat all-reduce) + optimizer step."""
        if self.reducer is not None:
            self.reducer.reduce()
        self.optimizer.step()

    def autocast(self):
        if self.mix_prec in (None, "f32"):
            return torch.autocast("cuda", enabled=False)
        return torch.autocast("cuda", dtype={"bf16": torch.bfloat16, "f16": torch.float16}[self.mix_prec])

    # --------------------------------------------------------------------------- paths / logs
    @property
    def save_path(self):
        p = os.path.join(self.path, "checkpoint")
        os.makedirs(p, exist_ok=True)
        return p

    @property
    def logs_path(self):
        p = os.path.join(self.path, "logs")
        os.makedirs(p, exist_ok=True)
        return p

    @property
    def network(self):
        return self.net

    @network.setter
    def network(self, new_val):
        self.net = new_val

    def write_log(self, log_str, prefix="valid", should_print=True):
        """prefix: valid, train, test (reference :232-249)"""
        if not self.is_root:
            return
        if prefix in ("valid", "test"):
            with open(os.path.join(self.logs_path, "valid_console.txt"), "a") as fout:
                fout.write(log_str + "\n")
        if prefix in ("valid", "test", "train"):
            with open(os.path.join(self.logs_path, "train_console.txt"), "a") as fout:
                if prefix in ("valid", "test"):
                    fout.write("=" * 10)
                fout.write(log_str + "\n")
        else:
            with open(os.path.join(self.logs_path, "%s.txt" % prefix), "a") as fout:
                fout.write(log_str + "\n")
        if should_print:
            print(log_str)

    # ------------------------------------------------------------------------- checkpoints
    def save_model(self, checkpoint=None, is_best=False, model_name=None):
        """checkpoint/{<model_name>,model_best}.pth.tar + latest.txt (reference :253-273)"""
        if not self.is_root:
            return
        if checkpoint is None:
            checkpoint = {"state_dict": self.network.state_dict()}
        if model_name is None:
            model_name = "checkpoint.pth.tar"
        checkpoint["dataset"] = self.run_config.dataset
        model_path = os.path.join(self.save_path, model_name)
        with open(os.path.join(self.save_path, "latest.txt"), "w") as fout:
            fout.write(model_path + "\n")
        torch.save(checkpoint, model_path)
        if is_best:
            torch.save({"state_dict": checkpoint["state_dict"]}, os.path.join(self.save_path, "model_best.pth.tar"))

    def load_model(self, model_fname=None):
        """reference :275-308 -- but failures are raised, not swallowed."""
        latest = os.path.join(self.save_path, "latest.txt")
        if model_fname is None and os.path.exists(latest):
            with open(latest) as fin:
                model_fname = fin.readline().rstrip("\n")
        if model_fname is None or not os.path.exists(model_fname):
            model_fname = "%s/checkpoint.pth.tar" % self.save_path
        checkpoint = torch.load(model_fname, map_location="cpu", weights_only=True)
        self.network.load_state_dict(checkpoint["state_dict"])
        if "epoch" in checkpoint:
            self.start_epoch = checkpoint["epoch"] + 1
        if "best_acc" in checkpoint:
            self.best_acc = checkpoint["best_acc"]
        if "optimizer" in checkpoint:
            self.optimizer.load_state_dict(checkpoint["optimizer"])
        return model_fname

    def save_config(self):
        if not self.is_root:
            return
        run_save_path = os.path.join(self.path, "run.config")
        with open(run_save_path, "w") as f:
            json.dump(self.run_config.config, f, indent=4)
        print("Run configs dump to %s" % run_save_path)

    # --------------------------------------------------------------------- validate / train
    def validate(self, epoch=0, is_test=True, run_str="", net=None, data_loader=None, no_logs=False,
                 tensorboard_logging=False, input_key="2x_down_image"):
        """eval-mode pass over the test (or valid) loader: (mean MSE loss, mean Y-PSNR).  The reference
        always feeds '2x_down_image' (:352-361, quirk Q4); `input_key` lifts that for 4x nets."""
        if net is None:
            net = self.net
        if data_loader is None:
            data_loader = self.run_config.test_loader if is_test else self.run_config.valid_loader
        net.eval()
        losses, psnrs = AverageMeter(), AverageMeter()
        with torch.no_grad():
            for mini_batch in data_loader:
                mini_batch = device_batch(mini_batch, self.device)
                images, lr = mini_batch["image"], mini_batch[input_key]
                with self.autocast():
                    output = net(lr)
                output = output.float()
                loss = self.test_criterion(output, images)
                losses.update(loss.item(), images.size(0))
                psnrs.update(psnr_y(output, images), images.size(0))
        return losses.avg, psnrs.avg

    def graphed(self, net):
        """the hipGraph-replayed forward of `net` in this manager's precision (graphed.GraphedEval), one per network"""
        cache = self.__dict__.setdefault("_graphed", {})
        g = cache.get(id(net))
        if g is None or g.net is not net:
            dt = None if self.mix_prec in (None, "f32") else {"bf16": torch.bfloat16, "f16": torch.float16}[self.mix_prec]
            g = cache[id(net)] = GraphedEval(net, autocast_dtype=dt)
        return g

    def validate_batched(self, net=None, data_loader=None, is_test=True, input_key="2x_down_image", max_batch=None,
                         graphs=None):
        """`validate` with the loader's batch-1 items of EQUAL size run as one batch (BASELINE config 5: Set14 at
        full resolution, reference eval_ofa_net_sr.py:187-220,247-251 / sr_run_manager.py:323-393).  Loss and PSNR are
        taken per image, so the result equals `validate` on the batch-1 loader (tests/test_hip_configs.py).
        `graphs` (default: on for 16-bit GPU inference unless OFASR_EVAL_GRAPHS=0): the forwards of all size buckets are
        captured once as ONE hipGraph and replayed -- the per-launch host cost otherwise bounds this loop (graphed.py).
        Returns (mean loss, mean PSNR, number of forward calls)."""
        if net is None:
            net = self.net
        if graphs is None:
            # 16-bit inference (the one-kernel blocks / ConvLayers) by default; the fp32 path is eager unless asked
            graphs = (os.environ.get("OFASR_EVAL_GRAPHS", "1") != "0" and str(self.device).startswith("cuda")
                      and self.mix_prec in ("bf16", "f16"))
        fwd = self.graphed(net) if graphs else None
        if data_loader is None:
            data_loader = self.run_config.test_loader if is_test else self.run_config.valid_loader
        net.eval()
        items = []
        for mini_batch in data_loader:
            mini_batch = device_batch(mini_batch, self.device)
            for i in range(mini_batch["image"].shape[0]):
                items.append({k: v[i:i + 1] for k, v in mini_batch.items() if torch.is_tensor(v)})
        losses, psnrs, calls = AverageMeter(), AverageMeter(), 0
        with torch.no_grad():
            groups = list(bucket_by_size(items, key=lambda it: it[input_key], max_batch=max_batch))
            lrs = [torch.cat([it[input_key] for it in group]).to(self.device) for group in groups]
            # all size buckets of the pass in ONE captured graph / one replay (the validation set is the same every epoch)
            outs = fwd.call_many(lrs) if (fwd is not None and lrs) else None
            for gi, group in enumerate(groups):
                images = torch.cat([it["image"] for it in group]).to(self.device)
                if outs is not None:
                    output = outs[gi].float()
                else:
                    with self.autocast():
                        output = net(lrs[gi]).float()
                calls += 1
                per_img = ((output - images) ** 2).mean(dim=(1, 2, 3))
                for v in per_img.tolist():
                    losses.update(v, 1)
                for v in psnr_y_per_image(output, images):
                    psnrs.update(v, 1)
        return losses.avg, psnrs.avg, calls

    def train_one_epoch(self, args, epoch, warmup_epochs=0, warmup_lr=0, input_key="2x_down_image"):
        """fixed-architecture ("teacher") epoch: BatchNorm layers run in eval mode (frozen statistics,
        reference :417-420), MSE loss, one optimizer step per batch.  Returns (mean loss, mean PSNR)."""
        self.net.train()
        for m in self.net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()
        loader = self.run_config.train_loader
        sampler = getattr(loader, "sampler", None)
        if hasattr(sampler, "set_epoch"):      # a rank-sharding sampler replays the same permutation otherwise
            sampler.set_epoch(epoch)
        nBatch = len(loader)
        losses, psnrs, data_time = AverageMeter(), AverageMeter(), AverageMeter()
        end = time.time()
        for i, mini_batch in enumerate(loader):
            data_time.update(time.time() - end)
            if epoch < warmup_epochs:
                new_lr = self.run_config.warmup_adjust_learning_rate(self.optimizer, warmup_epochs * nBatch, nBatch,
                                                                     epoch, i, warmup_lr)
            else:
                new_lr = self.run_config.adjust_learning_rate(self.optimizer, epoch - warmup_epochs, i, nBatch)
            mini_batch = device_batch(mini_batch, self.device)
            images, lr_img = mini_batch["image"], mini_batch[input_key]
            with self.autocast():
                output = self.net(lr_img)
            output = output.float()
            loss = self.train_criterion(output, images)
            teacher = getattr(args, "teacher_model", None)
            if teacher is not None:
                teacher.train()
                with torch.no_grad():
                    soft = teacher(images).detach()
                loss = args.kd_ratio * torch.nn.functional.mse_loss(output, soft) + loss
            self.zero_grad()
            loss.backward()
            ops.flush_deferred()   # the MB blocks' weight gradients: joined once per backward pass (ops.py)
            self.step()
            losses.update(loss.item(), images.size(0))
            psnrs.update(psnr_y(output, images), images.size(0))
            end = time.time()
        self._last_lr = new_lr
        return losses.avg, psnrs.avg

    def train(self, args, warmup_epoch=0, warmup_lr=0):
        """reference :516-541"""
        for epoch in range(self.start_epoch, self.run_config.n_epochs + warmup_epoch):
            train_loss, train_psnr = self.train_one_epoch(args, epoch, warmup_epoch, warmup_lr)
            is_best = False
            if (epoch + 1) % self.run_config.validation_frequency == 0:
                val_loss, val_psnr = self.validate(epoch=epoch, is_test=False)
                is_best = np.mean(val_psnr) > self.best_acc
                self.best_acc = max(self.best_acc, np.mean(val_psnr))
                self.write_log("Valid [%d/%d]\tloss %.3f\ttop-1 acc %.3f (%.3f)\tTrain top-1 %.3f\tloss %.3f\t" % (
                    epoch + 1 - warmup_epoch, self.run_config.n_epochs, np.mean(val_loss), np.mean(val_psnr),
                    self.best_acc, train_psnr, train_loss), prefix="valid", should_print=False)
            self.save_model({"epoch": epoch, "best_acc": self.best_acc, "optimizer": self.optimizer.state_dict(),
                             "state_dict": self.network.state_dict()}, is_best=is_best)

    def reset_running_statistics(self, net=None):
        from ...elastic_nn.utils import set_running_statistics
        set_running_statistics(self.network if net is None else net, self.run_config.train_loader)
