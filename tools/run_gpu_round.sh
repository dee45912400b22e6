cd $GRAFT_REPO_ROOT
for b in ubench_matrix_step_split ub_SMC_D ub_MC_DS ub_M_DCS ub_S_DMC ub__DSMC ubench_matrix_step_split; do echo $b; timeout 120 ./tools/ubench/$b | grep "nothing passes\|3.0 sigma" | grep -v "^pair"; done
