from .dist_autograd import DistAutogradAllGatherFunction  # noqa: F401
from .global_env import GlobalEnv, SummaryWriter  # noqa: F401
from .misc import DistSummaryWriter, seed_everything  # noqa: F401
