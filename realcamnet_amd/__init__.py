"""realcamnet_amd -- MI355X (gfx950) native RAW->sRGB learned-ISP inference path.

Host-side mirror of the reference interface (kepengxu/RealCamNet models/networks.py, models/LiteISP.py)
over the C ABI in include/realcam_hip.h (librealcam_hip.so, hand-written HIP).  No CPU fallback.
"""
from . import networks  # noqa: F401
from . import LiteISP  # noqa: F401
from . import groupmix  # noqa: F401
from . import tcm  # noqa: F401
from . import raw2bit  # noqa: F401
from . import graphs  # noqa: F401
from .graphs import GraphedCall  # noqa: F401
from .LiteISP import (ISPUNet_GFM, ISPUNet_GFM_crop, ISPUNet_GFM_LFM, ISPUNet_GFM_LSC, ISPUNet_GFM_LSC1, ISPUNet_GFM_LSC_noskip, ISPUNet_LSC, LiteISPNet, LiteISPNet_GFM, LiteISPNet_GFM_LSC,  # noqa: F401
                      LiteISPNet_GFM_LSC_GMA, LiteISPNet_GFMresize, LiteISPNet_LSC, ResUNet)
from .groupmix import GMA_Block  # noqa: F401

__all__ = ["networks", "LiteISP", "groupmix", "tcm", "raw2bit", "LiteISPNet", "LiteISPNet_GFM_LSC", "LiteISPNet_GFM_LSC_GMA", "LiteISPNet_LSC",
           "LiteISPNet_GFM", "LiteISPNet_GFMresize", "ISPUNet_GFM_LSC", "ISPUNet_GFM", "ISPUNet_GFM_LFM", "ISPUNet_LSC", "ResUNet", "ISPUNet_GFM_crop", "ISPUNet_GFM_LSC1",
           "ISPUNet_GFM_LSC_noskip", "GMA_Block", "graphs", "GraphedCall"]
