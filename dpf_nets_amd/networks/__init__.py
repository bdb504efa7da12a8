"""Host-side mirror of the reference's lib.networks surface for the per-point flow
decoder, the latent prior flow and the PointNet cloud encoder: same class names, constructor signatures, parameter/buffer names and
return structures (lib/networks/{layers,flows,decoders,losses,utils}.py)."""
from .layers import SharedDot, Swish  # noqa: F401
from .flows import CondRealNVPFlow3D, CondRealNVPFlow3DTriple  # noqa: F401
from .decoders import LocalCondRNVPDecoder  # noqa: F401
from .losses import PointFlowNLL, GaussianFlowNLL, GaussianEntropy, Local_Cond_RNVP_MC_Global_RNVP_VAE_Loss  # noqa: F401
from .encoders import PointNetCloudEncoder, FeatureEncoder, PointFeatures, TrainPointFeatures  # noqa: F401
from .optimizers import Adam, LRUpdater  # noqa: F401
from .prior_flows import RealNVPFlow, RealNVPFlowCouple, GlobalRNVPDecoder  # noqa: F401
from .models import Local_Cond_RNVP_MC_Global_RNVP_VAE  # noqa: F401
