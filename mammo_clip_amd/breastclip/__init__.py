"""Host-side mirror of the reference's ``breastclip`` package API for the contrastive pre-training hot path:
``model.build_model`` / ``loss.build_loss`` / ``util.GlobalEnv`` ... with identical names, argument meaning,
state_dict keys and error behaviour -- every tensor op underneath is a hand-written gfx950 kernel reached
through the C ABI (include/mammoclip_hip.h)."""
from . import loss, model, util  # noqa: F401
