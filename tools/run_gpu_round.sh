cd $GRAFT_REPO_ROOT
for rep in 1 2; do for b in ub_SM_DC ub_MS_DC ub_MC_DS ub_MC_SD ub_M_DSC ub_M_SDC ub_S_DMC ub_SMC_D ub_MSC_D ub_MCS_D ub_SM_CD; do echo -n "$b "; timeout 120 ./tools/ubench/$b | grep "3.0 sigma" | grep -v "^pair" | awk '{print $9}' ; done; done
