function x = quadprog_hip( H , f , A , b )
%quadprog_hip: the reference's solver shim (quadprog_gurobi.m:1, call sites Kmpc.m:382,809,882) on the device:
%   min 1/2 x'Hx + f'x  s.t.  A x <= b, dense strictly convex H.  Returns NaN(size(f)) when the QP is infeasible
%   (quadprog_gurobi.m:22-23).
    persistent owner        % KpOwner: the context reference is dropped by `clear quadprog_hip` / `clear all` / exit
    if isempty( owner ) || isempty( owner.value )
        owner = KpOwner( kp_mex( 'create' , 0 ) , 'destroy' );
    end
    x = kp_mex( 'qp_solve' , owner.value , full(H) , f(:) , full(A) , b(:) );
end
