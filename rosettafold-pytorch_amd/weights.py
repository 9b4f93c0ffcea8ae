"""Weight I/O between the reference's RoseTTAFold and this build.

The reference keeps two families of layers in plain Python lists, so their weights are NOT in its `state_dict()`
(`MsaUpdateWithPair.encoder_layers`, rf.py:602-605; `InitialCoordGenerationWithMsaAndPair.blocks`, rf.py:699-702).
Here they are registered `nn.ModuleList`s with the dotted names `<parent>.encoder_layers.<i>.*` / `<parent>.blocks.<i>.*`.
A reference checkpoint therefore comes in two parts:

    sd      = ref_model.state_dict()                    # registered parameters (same key names as this build)
    hidden  = export_hidden_lists(ref_model)            # the list-held layers, under this build's key names
    load_reference_weights(model, sd, hidden)           # strict: raises if anything this build needs is missing

`load_state_dict(sd, strict=False)` alone would silently leave the list-held layers at their random initialisation;
`load_reference_weights` refuses that unless told `allow_missing_hidden=True`.
"""
import torch

HIDDEN_LIST_ATTRS = ("encoder_layers", "blocks")


def export_hidden_lists(ref_model):
    """Weights of layers a (reference) model keeps in plain Python lists, keyed as this build's state_dict names them.
    Duck-typed: works on any nn.Module tree, needs nothing from the reference package."""
    out = {}
    for name, sub in ref_model.named_modules():
        for attr in HIDDEN_LIST_ATTRS:
            lst = getattr(sub, attr, None)
            if isinstance(lst, list):
                for i, layer in enumerate(lst):
                    for k, v in layer.state_dict().items():
                        out[f"{name + '.' if name else ''}{attr}.{i}.{k}"] = v.detach().clone()
    return out


def hidden_list_keys(model):
    """Keys of `model.state_dict()` that a reference `state_dict()` cannot contain (list-held layers)."""
    keys = []
    for k in model.state_dict():
        parts = k.split(".")
        if any(p in HIDDEN_LIST_ATTRS and i + 1 < len(parts) and parts[i + 1].isdigit() for i, p in enumerate(parts)):
            keys.append(k)
    return keys


def load_reference_weights(model, state_dict, hidden=None, allow_missing_hidden=False):
    """Load a reference checkpoint into `model` (this build's RoseTTAFold or any of its blocks).

    state_dict: the reference's `state_dict()` (may already contain the hidden-list keys); hidden: the result of
    `export_hidden_lists` (optional).  Every key of the model must be supplied with the right shape, except
      * the model's own non-reference buffers (none are persistent today), and
      * the list-held layers when `allow_missing_hidden=True` (they then keep their current values and are reported).
    Unknown keys in the inputs raise as well.  Returns {"loaded": n, "missing_hidden": [...]}.
    """
    merged = dict(state_dict)
    if hidden:
        clash = [k for k in hidden if k in merged and not torch.equal(merged[k].cpu(), hidden[k].cpu())]
        if clash:
            raise KeyError(f"hidden-list weights given twice with different values: {clash[:5]}")
        merged.update(hidden)
    own = model.state_dict()
    hid = set(hidden_list_keys(model))
    missing = [k for k in own if k not in merged]
    missing_hidden = [k for k in missing if k in hid]
    missing_other = [k for k in missing if k not in hid]
    unexpected = [k for k in merged if k not in own]
    if missing_other:
        raise KeyError(f"checkpoint lacks {len(missing_other)} registered keys, e.g. {missing_other[:5]}")
    if unexpected:
        raise KeyError(f"checkpoint holds {len(unexpected)} keys this model does not have, e.g. {unexpected[:5]}")
    if missing_hidden and not allow_missing_hidden:
        raise KeyError(
            f"{len(missing_hidden)} weights of list-held layers are missing (the reference's state_dict() cannot hold them: "
            f"rf.py:602-605, 699-702), e.g. {missing_hidden[:3]}; pass export_hidden_lists(ref_model) as `hidden`, or "
            "allow_missing_hidden=True to keep their current values")
    bad = [(k, tuple(merged[k].shape), tuple(own[k].shape)) for k in merged if tuple(merged[k].shape) != tuple(own[k].shape)]
    if bad:
        raise ValueError(f"shape mismatch for {len(bad)} keys, e.g. {bad[:3]}")
    res = model.load_state_dict(merged, strict=False)
    assert not res.unexpected_keys
    return {"loaded": len(merged), "missing_hidden": missing_hidden}


def save_checkpoint(model, path):
    """Full checkpoint of this build (registered + list-held layers): a plain `torch.save` of the state_dict."""
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)


def load_checkpoint(model, path):
    """Inverse of save_checkpoint; strict."""
    sd = torch.load(path, map_location="cpu")
    return load_reference_weights(model, sd, None, allow_missing_hidden=False)
