"""Hardware hand-off of a Solution: the numbers `LIFUInterface.set_solution`
(io/LIFUInterface.py:311-358) passes to `TxDevice.set_solution` (io/LIFUTXDevice.py:1317-1385), already in the
form the TX7332 profiles are built from.  Register addresses and bit positions are device tables and stay in
openlifu; serial transport, HV control and the device state machine are out of scope."""
from .tx_profile import DEFAULT_CLK_FREQ, DEFAULT_PATTERN_DUTY_CYCLE, DELAY_WIDTH, TxProfile, tx_profiles

__all__ = ["DEFAULT_CLK_FREQ", "DEFAULT_PATTERN_DUTY_CYCLE", "DELAY_WIDTH", "TxProfile", "tx_profiles"]
