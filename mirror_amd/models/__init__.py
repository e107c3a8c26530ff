"""Drop-in for the reference's `models` package (models/__init__.py:1-4) for the pre-training path."""
from .mirror import MIRROR, MIRRORClassifier, mirror, mirror_classifier, set_precision, resolve_precision  # noqa: F401

__all__ = ["mirror", "mirror_classifier"]

_REGISTRY = {"mirror": mirror, "mirror_classifier": mirror_classifier}


def create_model(model_name: str, pretrained: bool = False, checkpoint_path: str = "", scriptable=None, **kwargs):
    """Minimal stand-in for timm.models.create_model as called at train_mirror.py:689-694 (timm is optional)."""
    model = _REGISTRY[model_name](**kwargs)
    if checkpoint_path:
        from ..checkpoint import load_checkpoint_file
        state = load_checkpoint_file(checkpoint_path)
        model.load_state_dict(state.get("state_dict", state))
    return model


try:  # register with timm when it is installed so `timm.create_model("mirror")` resolves to this build
    from timm.models import register_model as _register_model

    _register_model(mirror)
    _register_model(mirror_classifier)
except Exception:  # timm absent: the local create_model above is the entry point
    pass
