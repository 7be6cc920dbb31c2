"""VGG input normalisation (reference model/losses/rgb_transform.py:5-21) as plain callables."""
from ...data.view_contract import post as _post
from ...data.view_contract import pre as _pre


def pre():
    """RGB [0,1] -> BGR, ImageNet-mean subtracted, x255. Returns a callable like the reference's Compose."""
    return _pre


def post():
    """Inverse of pre() + clamp to [0,1]; never mutates its input (the reference's in-place mul_ does on CPU)."""
    return _post
