"""gomilp_amd — MI355X-native dense-simplex LP-relaxation engine behind GoMILP's lp.Simplex boundary."""
