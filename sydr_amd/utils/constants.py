"""Constants the hot path and its host counterpart need (values as in sydr/utils/constants.py:4-6,71-85).

Two values of pi are in play in the reference and both are kept: NumPy's pi in
EPL / PCPS / the Borre NCO, and the GPS-ICD pi below in the Kaplan NCO and the discriminators."""
PI = 3.1415926535898            # GPS-ICD pi
HALF_PI = PI / 2.0
TWO_PI = PI * 2.0

GPS_L1CA_CODE_SIZE_BITS = 1023
GPS_L1CA_CODE_FREQ = 1.023e6
GPS_L1CA_CODE_MS = 1
GPS_L1CA_CARRIER_FREQ = 1575.42e6

LNAV_MS_PER_BIT = 20
LNAV_SUBFRAME_SIZE = 300
LNAV_WORD_SIZE = 30

# Kaplan digital loop filter constants ([Kaplan, 2006] p.180)
W0_BANDWIDTH_1 = 0.25
W0_BANDWIDTH_2 = 0.53
W0_BANDWIDTH_3 = 0.7845
W0_SCALE_A2 = 1.414
W0_SCALE_A3 = 1.1
W0_SCALE_B3 = 2.4
