cd $GRAFT_REPO_ROOT
for b in ubench_matrix_step ubench_matrix_step_inline; do echo $b; timeout 120 ./tools/ubench/$b | grep -v "^pair"; done
