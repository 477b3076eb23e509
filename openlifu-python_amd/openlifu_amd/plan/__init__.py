from __future__ import annotations

from .protocol import OnPulseMismatchAction, Protocol
from .solution import Solution
from .solution_analysis import SolutionAnalysis, SolutionAnalysisOptions, get_focus_matrix
from .target_constraints import TargetConstraints

__all__ = ["Protocol", "Solution", "SolutionAnalysis", "SolutionAnalysisOptions", "TargetConstraints",
           "OnPulseMismatchAction", "get_focus_matrix"]
