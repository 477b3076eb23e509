"""Planning API boundary: ``Protocol`` (beamform / calc_solution) and ``Solution`` (scale / analyze)."""
from __future__ import annotations

from . import param_constraint as _param
from . import protocol as _protocol
from . import solution as _solution
from . import solution_analysis as _analysis
from . import target_constraints as _constraints

Protocol = _protocol.Protocol
OnPulseMismatchAction = _protocol.OnPulseMismatchAction
Solution = _solution.Solution
SolutionAnalysis = _analysis.SolutionAnalysis
SolutionAnalysisOptions = _analysis.SolutionAnalysisOptions
get_focus_matrix = _analysis.get_focus_matrix
get_offset_grid = _analysis.get_offset_grid
get_gridded_transformed_coords = _analysis.get_gridded_transformed_coords
calc_dist_from_focus = _analysis.calc_dist_from_focus
get_mask = _analysis.get_mask
TargetConstraints = _constraints.TargetConstraints
ParameterConstraint = _param.ParameterConstraint

__all__ = ("Protocol", "Solution", "SolutionAnalysis", "SolutionAnalysisOptions", "TargetConstraints",
           "OnPulseMismatchAction", "get_focus_matrix", "get_offset_grid", "get_gridded_transformed_coords",
           "calc_dist_from_focus", "get_mask", "ParameterConstraint")
