"""The parity bar for f64 outputs that pass through a device transcendental (atan / sin / pow / log): |g - e| <= 1e-12 *
max(|e|, scale).  scale = 0 is the plain relative bound of BASELINE.json's north_star.  Outputs that cross zero cannot be held to a
relative bound AT the crossing and are judged against their natural scale -- only these, measured in profiles/r03_tolerance.json
(worst scaled error 1.3e-14):
  ht_dcphase            an angle in degrees over (-45, 315): one turn
  inphase / quadrature  detrended price components, i.e. differences of O(price) terms: the price level of the row ("price")
  sine / leadsine       sines: 1
Every other such output (ht_dcperiod, mama, fama) keeps scale 0.  Everything that does not pass through a transcendental is
compared bit for bit."""
RTOL = 1e-12
SCALE_OF = {"ht_dcphase.ht_dcphase": 360.0, "ht_phasor.inphase": "price", "ht_phasor.quadrature": "price", "ht_sine.sine": 1.0,
            "ht_sine.leadsine": 1.0}
