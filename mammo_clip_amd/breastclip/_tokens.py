"""Token batches arrive as a HF ``BatchEncoding`` (has .to) or a plain dict of tensors."""


def to_device(tokens, device):
    if hasattr(tokens, "to") and not isinstance(tokens, dict):
        return tokens.to(device)
    if hasattr(tokens, "to") and type(tokens) is not dict:
        try:
            return tokens.to(device)
        except TypeError:
            pass
    return {k: (v.to(device) if hasattr(v, "to") else v) for k, v in tokens.items()}
