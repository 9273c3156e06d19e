"""Drop-in module paths: the names the reference's callers import (`src.prediction.models.dynamics`,
`src.cem.cem`, ...) resolved to the MI355X implementation in `robot_aware_control_amd`."""
