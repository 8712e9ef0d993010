"""Counterpart of the reference's `processors/ddp_pose_resnet_solver.py` (`DDPProcessor`, :20-212) for the hot path: same yaml
schema, one process per GPU (env:// rendezvous, `torch.distributed` backend nccl = RCCL), every per-iteration operation a
HIP kernel behind `simple_pose_amd` (train step, HeatMapAcc, decoder).

Out of scope by the survey's contract: COCO parsing / augmentation and COCOeval.  The loaders are
therefore pluggable (`train_loader` / `val_loader`: iterables of the reference's collate tuples, tensors on any device), and
`data.synthetic: N` builds deterministic synthetic ones on the GPU; `val()` reports loss / accuracy and writes the
checkpoint in the reference's format, and computes AP only when pycocotools is importable and an annotation file is given."""
from __future__ import annotations

import json
import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist
import yaml

from .. import synth
from ..commons.transforms import RefineSimpleTransform
from ..metrics.pose_metrics import BasicKeyPointDecoder, HeatMapAcc, kps_to_dict_
from ..nets import pose_resnet_dconv, pose_resnet_duc
from ..sharding import rank_indices
from ..train import PoseTrainer

_MODEL_TYPES = {"pose_resnet_dconv": pose_resnet_dconv, "pose_resnet_duc": pose_resnet_duc}


class AverageLogger(object):
    """commons/model_utils.py:93-110, accumulating DEVICE scalars: no `.item()` per iteration (ddp...:132-133 syncs twice each)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.data, self.count = None, 0

    def update(self, value: torch.Tensor):
        self.data = value.detach().clone() if self.data is None else self.data + value
        self.count += 1

    def avg(self) -> float:
        return float(self.data.item()) / self.count if self.count else 0.0


def multi_step_lr(base_lr: float, milestones, gamma: float, epoch: int) -> float:
    """torch.optim.lr_scheduler.MultiStepLR (ddp...:73-77) in closed form: lr during epoch `epoch` (0-based)."""
    return base_lr * gamma ** sum(1 for m in milestones if epoch >= m)


class SyntheticLoader:
    """`n` images per epoch for this rank as collate tuples (input [B,3,256,192], heat maps [B,J,64,48], masks [B,J], trans_inv
    [B,2,3], ids), generated once and kept in HBM; sample i of the global epoch goes to rank i % world (DistributedSampler)."""

    def __init__(self, n: int, batch_size: int, joints: int, rank: int, world: int, device, seed: int = 512):
        idx = rank_indices(n * world, rank, world, batch_size=batch_size)
        self.batches = []
        for b0 in range(0, len(idx), batch_size):
            ids = idx[b0:b0 + batch_size]
            if len(ids) < batch_size:
                break                                        # drop_last=True (ddp...:43-44)
            x = torch.from_numpy(synth.input_images(len(ids), seed + ids[0])).to(device)
            j = torch.from_numpy(synth.joints_batch(len(ids), joints, seed=seed + 7 + ids[0])).to(device)
            hm, mask = RefineSimpleTransform.get_heat_map(j, 2.0, (48, 64))          # HIP encoder
            tinv = torch.from_numpy(synth.trans_inv_batch(len(ids))).to(device)
            self.batches.append((x, hm, mask, tinv, [int(i) for i in ids]))

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


class DDPProcessor(object):
    def __init__(self, cfg_path: str, train_loader: Optional[Iterable] = None, val_loader: Optional[Iterable] = None):
        with open(cfg_path, "r") as rf:
            self.cfg = yaml.safe_load(rf)
        self.data_cfg, self.model_cfg = self.cfg["data"], self.cfg["model"]
        self.optim_cfg, self.val_cfg = self.cfg["optim"], self.cfg["val"]
        if not dist.is_initialized():                         # launched by torch.distributed.run: env:// rendezvous (ddp...:36)
            if "RANK" in os.environ:
                dist.init_process_group(backend=os.environ.get("SP_DIST_BACKEND", "nccl"))
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.local_rank = int(os.environ.get("LOCAL_RANK", 0)) % max(1, torch.cuda.device_count())
        self.device = torch.device("cuda", self.local_rank)
        torch.cuda.set_device(self.device)
        if self.model_cfg["type"] not in _MODEL_TYPES:
            raise ValueError(f"model.type {self.model_cfg['type']!r}: expected one of {sorted(_MODEL_TYPES)}")
        model = getattr(_MODEL_TYPES[self.model_cfg["type"]], self.model_cfg["name"])(
            pretrained=self.model_cfg["pretrained"], num_classes=self.model_cfg["num_joints"])
        self.model = model.to(self.device).train()
        self.base_lr = float(self.optim_cfg["lr"])
        # N > 1 over RCCL: the step's collectives go to RCCL directly when a start-up self-check (a few 4-image steps through both paths,
        # compared bit for bit on every rank) says the native path reproduces torch.distributed; else torch.distributed, reason logged
        from .. import comm_select
        dtype = "bf16" if self.optim_cfg["amp"] else "fp32"
        self.collectives = comm_select.select(self.model, None, 256, 192, dtype, bool(self.optim_cfg["sync_bn"]))
        self.trainer = PoseTrainer(self.model, lr=self.base_lr, dtype=dtype, sync_bn=bool(self.optim_cfg["sync_bn"]),
                                   native_comm=self.collectives["native"] if self.world > 1 else None)
        bs, J = int(self.data_cfg["batch_size"]), int(self.model_cfg["num_joints"])
        n_syn = self.data_cfg.get("synthetic")
        if train_loader is None:
            if not n_syn:
                raise NotImplementedError("COCO loading is outside the hot path (SURVEY.md section 2): pass train_loader / val_loader "
                                          "or set data.synthetic")
            train_loader = SyntheticLoader(int(n_syn), bs, J, self.rank, self.world, self.device)
        if val_loader is None and n_syn:
            val_loader = SyntheticLoader(max(bs, int(n_syn) // 4), bs, J, 0, 1, self.device, seed=9001)
        self.tloader, self.vloader = train_loader, val_loader
        self.acc_func = HeatMapAcc()
        self.loss_logger, self.acc_logger = AverageLogger(), AverageLogger()
        self.decoder = BasicKeyPointDecoder()
        self.best_map = 0.0
        self.history = []

    # ddp...:97-153
    def train(self, epoch: int):
        self.model.train()
        self.loss_logger.reset(); self.acc_logger.reset()
        self.trainer.lr = multi_step_lr(self.base_lr, self.optim_cfg["milestones"], self.optim_cfg["gamma"], epoch)
        for input_tensors, heat_maps, masks, _, _ in self.tloader:
            x = input_tensors.to(self.device, non_blocking=True)
            targets = heat_maps.to(self.device, non_blocking=True)
            mask = masks.to(self.device, non_blocking=True)
            if self.trainer.tuned_for_batch == 0 and x.shape[0] >= 8:
                # once: fastest conv tile per layer at this per-GPU batch, timed on rank 0 and shared (the BatchNorm partial sums are
                # grouped per tile, so a per-rank choice would let the ranks' statistics differ by rounding)
                self.trainer.autotune_shared(x.shape[0])
            loss = self.trainer.step(x, targets, mask)        # zero_grad / forward / loss / backward / all-reduce / Adam
            acc = self.acc_func(self.trainer.last_heat, targets, mask)
            self.loss_logger.update(loss[0]); self.acc_logger.update(acc)
        mean_loss, mean_acc = self._reduce_mean(self.loss_logger.avg()), self._reduce_mean(self.acc_logger.avg())
        self.history.append({"epoch": epoch, "lr": self.trainer.lr, "loss": mean_loss, "acc": mean_acc})
        if self.rank == 0:
            print("train epoch:{:d}|mean_loss:{:8.6f}|mean_acc:{:6.4f}|lr:{:8.6f}".format(epoch + 1, mean_loss, mean_acc * 100, self.trainer.lr))

    def _reduce_mean(self, v: float) -> float:                # reduce_sum(...) / gpu_num, ddp...:144-145
        if self.world == 1:
            return v
        t = torch.tensor(v, device=self.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item()) / self.world

    # ddp...:155-206
    @torch.no_grad()
    def val(self, epoch: int):
        if self.rank != 0 or self.vloader is None:
            return None
        from .. import _lib
        self.loss_logger.reset(); self.acc_logger.reset()
        self.model.eval()
        kps_dict_list = []
        ws = torch.empty(4096, dtype=torch.uint8, device=self.device)
        for input_tensors, heat_maps, masks, trans_invs, img_ids in self.vloader:
            x, targets = input_tensors.to(self.device), heat_maps.to(self.device)
            tinv, mask = trans_invs.to(self.device).float(), masks.to(self.device)
            predicts = self.model(x)
            loss = torch.zeros(1, dtype=torch.float32, device=self.device)
            B, J, H, W = predicts.shape
            _lib.check(_lib.lib().sp_masked_mse(_lib.ptr(predicts), _lib.ptr(targets.contiguous()), _lib.ptr(mask.contiguous()), B, J, H * W,
                                                _lib.ptr(loss), None, _lib.ptr(ws), _lib.current_stream()), "sp_masked_mse")
            acc = self.acc_func(predicts, targets, mask)
            pred_kps, scores = self.decoder(predicts, tinv)
            kps_to_dict_(pred_kps, scores, img_ids, kps_dict_list)
            self.loss_logger.update(loss[0]); self.acc_logger.update(acc)
        self.model.train()
        cpkt = {"ema": {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}, "epoch": epoch}
        val_ap = self._evaluate_map(kps_dict_list)
        out = {"epoch": epoch, "loss": self.loss_logger.avg(), "acc": self.acc_logger.avg(), "val_ap": val_ap, "results": len(kps_dict_list)}
        print("val epoch:{:d}|mean_loss:{:8.6f}|mean_acc:{:6.4f}|val_ap:{}".format(epoch + 1, out["loss"], out["acc"] * 100, val_ap))
        wdir = self.val_cfg["weight_path"]
        os.makedirs(wdir, exist_ok=True)
        if val_ap is not None and val_ap > self.best_map:
            self.best_map = val_ap
            torch.save(cpkt, os.path.join(wdir, "{:s}_best.pth".format(self.cfg["model_name"])))
        torch.save(cpkt, os.path.join(wdir, "{:s}_last.pth".format(self.cfg["model_name"])))
        return out

    def _evaluate_map(self, kps_dict_list):
        ann = self.data_cfg.get("val_ann_path")
        if not ann or not os.path.isfile(ann):
            return None                                       # synthetic data has no ground-truth annotation file
        try:
            from pycocotools.coco import COCO
            from pycocotools.cocoeval import COCOeval
        except ImportError:
            return None
        with open("temp_test.json", "w") as wf:
            json.dump(kps_dict_list, wf)
        gt = COCO(ann)
        ev = COCOeval(gt, gt.loadRes("temp_test.json"), "keypoints")
        ev.evaluate(); ev.accumulate(); ev.summarize()
        return float(ev.stats[0])

    def run(self):
        for epoch in range(self.optim_cfg["epochs"]):
            self.train(epoch)
            if (epoch + 1) % self.val_cfg["interval"] == 0:
                self.val(epoch)


if __name__ == "__main__":                                    # main.py of the reference: DDPProcessor(cfg_path=...).run()
    import sys
    DDPProcessor(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                      "configs", "ddp_fast_pose.yaml")).run()
    if dist.is_initialized():
        dist.destroy_process_group()
