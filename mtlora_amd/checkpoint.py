"""Checkpoint interop for the MI355X MTLoRA modules: vanilla Swin / reference MTLoRA checkpoints into
``mtlora_amd.swin_transformer_mtlora`` models, and LoRA merging for inference.

Mirrors reference ``utils.py:41-176`` (``load_checkpoint``: key mapping :125-149, ``attn_mask`` strip :60-63,
relative-position table / absolute position embedding re-interpolation :65-121) and ``models/lora.py:636-668``
(``merge_lora_weights``, ``map_old_state_dict_weights``) -- same arguments, same side effects on the state dict, same
``strict=False`` load -- so a maintainer can point the reference's ``main.py`` at it unchanged.  Pure host code.
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Mapping, MutableMapping, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .lora import MTLoRALinear, map_old_state_dict_weights

_LOG = logging.getLogger("mtlora_amd.checkpoint")


def _get(ns: Any, name: str, default=None):
    if ns is None:
        return default
    if isinstance(ns, Mapping):
        return ns.get(name, default)
    return getattr(ns, name, default)


def lora_key_mapping(model_state: Mapping[str, torch.Tensor], mtlora: Any) -> Dict[str, str]:
    """``{old key: new key}`` for the linears that the MTLoRA config turned into ``MTLoRALinear`` (their parameters live
    under ``.linear.``): reference utils.py:125-145."""
    layers: List[str] = []
    if _get(mtlora, "QKV_ENABLED"):
        layers += ["attn.qkv.weight", "attn.qkv.bias"]
    if _get(mtlora, "PROJ_ENABLED"):
        layers += ["attn.proj.weight", "attn.proj.bias"]
    if _get(mtlora, "FC1_ENABLED"):
        layers += ["mlp.fc1.weight", "mlp.fc1.bias"]
    if _get(mtlora, "FC2_ENABLED"):
        layers += ["mlp.fc2.weight", "mlp.fc2.bias"]
    if _get(mtlora, "DOWNSAMPLER_ENABLED"):
        layers += ["downsample.reduction.weight"]
    mapping = {}
    for k in model_state:
        parts = k.split(".")
        last_three, prefix = ".".join(parts[-3:]), ".".join(parts[:-3])
        if last_three in layers:
            wb = parts[-1]
            layer = ".".join(parts[-3:-1])
            mapping[f"{prefix}.{layer}.{wb}"] = f"{prefix}.{layer}.linear.{wb}"
    return mapping


def prepare_state_dict(model: nn.Module, model_state: MutableMapping[str, torch.Tensor], mtlora: Any = None,
                       update_relative_position: bool = False, skip_decoder: bool = False, split_qkv: bool = False,
                       logger: Optional[logging.Logger] = None) -> MutableMapping[str, torch.Tensor]:
    """everything reference ``load_checkpoint`` does to ``checkpoint["model"]`` before ``load_state_dict`` (in place)."""
    log = logger or _LOG
    if skip_decoder:  # utils.py:55-56
        for k in [k for k in model_state if k.startswith("decoders")]:
            del model_state[k]
    for k in [k for k in model_state if "attn_mask" in k]:  # re-initialised by the constructor (utils.py:59-62)
        del model_state[k]
    if update_relative_position:
        for pat in ("relative_position_index", "relative_coords_table"):  # utils.py:65-76
            for k in [k for k in model_state if pat in k]:
                del model_state[k]
        current = model.state_dict()
        for k in [k for k in model_state if "relative_position_bias_table" in k]:  # utils.py:78-98
            pre = model_state[k]
            if k not in current:
                continue
            L1, nH1 = pre.shape
            L2, nH2 = current[k].shape
            if nH1 != nH2:
                log.warning(f"Error in loading {k}, passing......")
            elif L1 != L2:
                S1, S2 = int(L1 ** 0.5), int(L2 ** 0.5)
                r = F.interpolate(pre.permute(1, 0).view(1, nH1, S1, S1), size=(S2, S2), mode="bicubic")
                model_state[k] = r.view(nH2, L2).permute(1, 0)
        for k in [k for k in model_state if "absolute_pos_embed" in k]:  # utils.py:100-121
            pre = model_state[k]
            if k not in current:
                continue
            _, L1, C1 = pre.shape
            _, L2, _ = current[k].shape
            if L1 != L2:
                S1, S2 = int(L1 ** 0.5), int(L2 ** 0.5)
                r = F.interpolate(pre.reshape(-1, S1, S1, C1).permute(0, 3, 1, 2), size=(S2, S2), mode="bicubic")
                model_state[k] = r.permute(0, 2, 3, 1).flatten(1, 2)
    if _get(mtlora, "ENABLED"):  # utils.py:123-149
        mapping = lora_key_mapping(model_state, mtlora)
        if not mapping:
            print("No keys needs to be mapped for LoRA")
        map_old_state_dict_weights(model_state, mapping, "", split_qkv)
    return model_state


def load_state(model: nn.Module, model_state: MutableMapping[str, torch.Tensor], mtlora: Any = None,
               update_relative_position: bool = False, skip_decoder: bool = False, split_qkv: bool = False,
               logger: Optional[logging.Logger] = None, quiet: bool = False) -> Tuple[List[str], List[str]]:
    """prepare + ``load_state_dict(strict=False)``; returns (missing, unexpected) and logs them like the reference."""
    log = logger or _LOG
    prepare_state_dict(model, model_state, mtlora, update_relative_position, skip_decoder, split_qkv, log)
    res = model.load_state_dict(model_state, strict=False)
    missing, unexpected = list(res.missing_keys), list(res.unexpected_keys)
    if not quiet:
        if missing:
            log.warning("=============Missing Keys==============")
            for k in missing:
                log.warning(k)
        if unexpected:
            log.warning("=============Unexpected Keys==============")
            for k in unexpected:
                log.warning(k)
    return missing, unexpected


def load_checkpoint(config, model, optimizer, lr_scheduler, loss_scaler, logger, backbone: bool = False, quiet: bool = False):
    """Drop-in for reference ``utils.load_checkpoint`` (utils.py:41-176): same signature, same config fields read
    (``MODEL.RESUME[_BACKBONE]``, ``MODEL.MTLORA``, ``MODEL.UPDATE_RELATIVE_POSITION``, ``TRAIN.SKIP_DECODER_CKPT``,
    ``EVAL_MODE``), returns ``max_accuracy``."""
    path = config.MODEL.RESUME if not backbone else config.MODEL.RESUME_BACKBONE
    logger.info(f"==============> Resuming form {path}....................")
    if str(path).startswith("https"):
        ckpt = torch.hub.load_state_dict_from_url(path, map_location="cpu", check_hash=True)
    else:
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    mtlora = config.MODEL.MTLORA
    skip_decoder = bool(_get(config.TRAIN, "SKIP_DECODER_CKPT", False))
    load_state(model, ckpt["model"], mtlora, bool(_get(config.MODEL, "UPDATE_RELATIVE_POSITION", False)), skip_decoder,
               bool(_get(mtlora, "SPLIT_QKV", False)), logger, quiet)
    max_accuracy = 0.0
    if (not _get(config, "EVAL_MODE", False) and "optimizer" in ckpt and "lr_scheduler" in ckpt and "epoch" in ckpt
            and not skip_decoder):
        optimizer.load_state_dict(ckpt["optimizer"])
        lr_scheduler.load_state_dict(ckpt["lr_scheduler"])
        if hasattr(config, "defrost"):
            config.defrost()
        config.TRAIN.START_EPOCH = ckpt["epoch"] + 1
        if hasattr(config, "freeze"):
            config.freeze()
        if "scaler" in ckpt and loss_scaler is not None:
            loss_scaler.load_state_dict(ckpt["scaler"])
        logger.info(f"=> loaded successfully '{path}' (epoch {ckpt['epoch']})")
        max_accuracy = ckpt.get("max_accuracy", 0.0)
    del ckpt
    return max_accuracy


def merge_lora_weights(model: nn.Module) -> int:
    """Fold the low-rank updates into the frozen weights for inference (the role of reference lora.py:636-641, whose
    ``MTLoRALinear.merge`` is a stub): calls ``merge()`` on every ``MTLoRALinear``; returns how many layers merged.  Layers
    whose task outputs do not see the shared update (``shared_mode='matrix'`` with tasks) cannot be expressed by one weight
    and stay as they are."""
    return sum(1 for m in model.modules() if isinstance(m, MTLoRALinear) and m.merge())


def unmerge_lora_weights(model: nn.Module) -> int:
    return sum(1 for m in model.modules() if isinstance(m, MTLoRALinear) and m.unmerge())
