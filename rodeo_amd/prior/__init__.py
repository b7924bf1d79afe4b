from .ibm import ibm_init, ibm_state
from .indep_init import indep_init
