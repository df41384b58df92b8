function x = quadprog_hip( H , f , A , b )
%quadprog_hip: the reference's solver shim (quadprog_gurobi.m:1, call sites Kmpc.m:382,809,882) on the device:
%   min 1/2 x'Hx + f'x  s.t.  A x <= b, dense strictly convex H.  Returns NaN(size(f)) when the QP is infeasible
%   (quadprog_gurobi.m:22-23).
    persistent ctx
    if isempty( ctx )
        ctx = kp_mex( 'create' , 0 );
    end
    x = kp_mex( 'qp_solve' , ctx , full(H) , f(:) , full(A) , b(:) );
end
