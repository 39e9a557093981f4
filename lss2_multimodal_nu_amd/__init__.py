"""MI355X-native Lift-Splat-Shoot camera->BEV path (drop-in for the hot path of
fircarpediem/LSS2_Multimodal_nu).  Importing the package does not touch the GPU;
the HIP library is loaded on first use and its absence is an error."""
from .tools import MultiLoss, QuickCumsum, SimpleLoss, cumsum_trick, gen_dx_bx  # noqa: F401
from .data import CalibrationPack, prepare_calibration  # noqa: F401
from .modules import BevEncode, CamEncode, Encoder, Up, enable_sync_bn  # noqa: F401
from .model_BEV_TXT import BEV_TXT, LSS, compile_model_bevtxt, compile_model_lss  # noqa: F401
from .model_baseline import compile_model_onlybev  # noqa: F401  (its BEV_TXT: model_baseline.BEV_TXT)
from .model_vovnet_transformer import (BEVEncoderTransformer, CamEncodeV2, MultiScaleDepthNet,  # noqa: F401
                                       StandardDepthNet, VoVNetBEVTransformer,
                                       compile_model_vovnet_transformer)
from .transformer_modules import LightweightBEVTransformer  # noqa: F401
from .optim import ClipAdam  # noqa: F401

__all__ = ["ClipAdam", "gen_dx_bx", "cumsum_trick", "QuickCumsum", "SimpleLoss", "MultiLoss", "CalibrationPack",
           "prepare_calibration", "enable_sync_bn", "Up", "Encoder", "CamEncode", "BevEncode", "LSS", "BEV_TXT",
           "compile_model_lss", "compile_model_bevtxt", "compile_model_onlybev", "StandardDepthNet", "MultiScaleDepthNet", "CamEncodeV2",
           "BEVEncoderTransformer", "LightweightBEVTransformer", "VoVNetBEVTransformer",
           "compile_model_vovnet_transformer"]
