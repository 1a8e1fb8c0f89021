from robo_amd.acquisition_functions.base_acquisition import BaseAcquisitionFunction  # noqa: F401
from robo_amd.acquisition_functions.ei import EI  # noqa: F401
from robo_amd.acquisition_functions.log_ei import LogEI  # noqa: F401
from robo_amd.acquisition_functions.pi import PI  # noqa: F401
from robo_amd.acquisition_functions.lcb import LCB  # noqa: F401
from robo_amd.acquisition_functions.marginalization import MarginalizationGPMCMC  # noqa: F401
from robo_amd.acquisition_functions.information_gain import InformationGain  # noqa: F401
from robo_amd.acquisition_functions.information_gain_per_unit_cost import InformationGainPerUnitCost  # noqa: F401
