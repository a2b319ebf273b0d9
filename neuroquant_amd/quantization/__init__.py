from .quant_block import BaseQuantBlock, QuantNeRVBlock
from .quant_layer import QuantModule
from .quant_model import QuantModel
from .quantizer import AdaRoundQuantizer, UniformAffineQuantizer, lp_loss
from .data_utils import LinearTempDecay
from .calib_model import LossFunction, model_reconstruction
