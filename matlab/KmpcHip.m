classdef KmpcHip < Kmpc
    %KmpcHip: the reference's Kmpc with the per-step work on an MI355X (one kernel launch per MPC step).
    %   Drop-in: replace `Kmpc(` by `KmpcHip(` in example_control.m / Kmpc_setup.m (sysid_class must be a KsysidHip);
    %   Ksim.run_trial_mpc calls get_mpcInput / get_mpcInput_bilinear_iter exactly as before (Ksim.m:206-216) and still
    %   tests any(isnan(U)) for solver failure (:220).  Overridden: the three per-step entry points (Kmpc.m:329-387,
    %   750-814, 817-904: lift, get_costB/H/G/D_bilinear, constraint rows, quadprog) and, for loaded models, the load
    %   estimators (Kmpc.m:1298-1445: regression assembled on the host, lsqlin replaced by the device QP).  The constructor
    %   of the parent still assembles its own cost / constraint matrices (used by nothing below, kept for inspection).
    %   mpc_type 'nonlinear' (fmincon) stays the parent's.

    properties
        hip;    % struct: mpc (uint64 handle), sys (the KsysidHip)
    end

    methods
        function obj = KmpcHip( sysid_class , varargin )
            obj = obj@Kmpc( sysid_class , varargin{:} );
            if ~isa( sysid_class , 'KsysidHip' )
                error( 'KmpcHip needs a KsysidHip (it shares its device context and dictionary)' );
            end
            obj.hip.sys = sysid_class;
            if strcmp( obj.mpc_type , 'nonlinear' )
                return;
            end
            m = obj.params.m;
            r = obj.cost_input(:);                       % eye(m).*cost_input (Kmpc.m:201,548): a column vector scales rows
            if numel( r ) == 1
                r = r * ones( m , 1 );
            end
            lo = []; hi = [];
            if ~isempty( obj.input_bounds )              % scaled-down bounds (:247,659)
                sc = obj.scaledown.u( obj.input_bounds' )';
                lo = sc(:,1); hi = sc(:,2);
            end
            slope = [];
            if ~isempty( obj.input_slopeConst )
                slope = obj.input_slopeConst * mean( obj.params.scale.u_factor );            % :272,684
            end
            smooth = [];
            if ~isempty( obj.input_smoothConst )
                smooth = obj.params.Ts^2 * obj.input_smoothConst * mean( obj.params.scale.u_factor );   % :294,706
            end
            mt = double( strcmp( obj.model_type , 'bilinear' ) );
            % loaded models: A, B already have the loaded sizes N (nw+1) (Ksysid.m:1192-1200, :1251-1259)
            obj.hip.mpc = kp_mex( 'mpc_create' , sysid_class.hip.ctx , mt , obj.model.A , obj.model.B , obj.horizon , ...
                                  obj.projmtx , obj.cost_running , obj.cost_terminal , r , lo , hi , slope , smooth );
            % released when the last copy of this (value-class) object is gone; the controller points into the context
            obj.hip.mpc_owner = KpOwner( obj.hip.mpc , 'mpc_destroy' , sysid_class.hip.ctx_owner );
            if ~isempty( obj.state_bounds )              % :313
                sb = obj.scaledown.y( obj.state_bounds' )';
                kp_mex( 'mpc_set_state_bounds' , obj.hip.mpc , sb(:,1) , sb(:,2) );
            end
            if obj.loaded                                % the parent's lift handle, served by the device (Ksysid.m:1606-1612)
                obj.lift.econ_full_loaded = sysid_class.lift.econ_full_loaded;
            end
        end

        function [ U , z ] = hip_step( obj , traj , ref , iters )
            [ ~ , zeta_all ] = obj.get_zeta( traj );     % Kmpc.m:343-344
            zeta = zeta_all( end , : )';
            Np = obj.horizon;                            % reference padding (:354-365)
            if size( ref , 2 ) ~= size( obj.projmtx , 1 )
                error( 'Reference trajectory is not the correct dimension' );
            elseif size( ref , 1 ) > Np + 1
                ref = ref( 1 : Np + 1 , : );
            elseif size( ref , 1 ) < Np + 1
                ref = [ ref ; kron( ones( Np + 1 - size(ref,1) , 1 ) , ref(end,:) ) ];
            end
            Yr = reshape( ref' , [] , 1 );
            if obj.loaded                                % :347-348, :771-772, :839-840: lift with the current load estimate
                z = obj.lift.econ_full_loaded( zeta , traj.what( end , : )' );
                U = kp_mex( 'mpc_step' , obj.hip.mpc , z , traj.u(end,:)' , Yr , iters );
                return;
            end
            [ U , z ] = kp_mex( 'mpc_step_zeta' , obj.hip.mpc , obj.hip.sys.hip.basis , zeta , traj.u(end,:)' , Yr , iters );
        end

        function [ U , status ] = hip_step_batch( obj , Z , Uprev , refs )
            % nb independent problems of this controller in one launch (Monte-Carlo closed loops, random-state sweeps;
            % the reference would call get_mpcInput* nb times, Kmpc.m:329, :750, :817).  Z: N x nb lifted states,
            % Uprev: m x nb previous inputs, refs: (Np+1) x nproj x nb reference windows (already padded, :354-365).
            % U: Np x m x nb (row 1 = pinned current input); NaN where the QP failed (status(i) ~= 0)
            nb = size( Z , 2 );
            YR = reshape( permute( refs , [ 2 1 3 ] ) , [] , nb );       % Yr = reshape( ref' , [] , 1 ) per problem (:858)
            [ U , status ] = kp_mex( 'mpc_step_batch' , obj.hip.mpc , Z , Uprev , YR );
        end

        function [ U , z ] = get_mpcInput( obj , traj , ref )
            [ U , z ] = obj.hip_step( traj , ref , 1 );                % Kmpc.m:329-387
        end

        function [ U , z ] = get_mpcInput_bilinear( obj , traj , ref )
            [ U , z ] = obj.hip_step( traj , ref , 1 );                % Kmpc.m:750-814
        end

        function [ U , z ] = get_mpcInput_bilinear_iter( obj , traj , ref , iter )
            [ U , z ] = obj.hip_step( traj , ref , iter );             % Kmpc.m:817-904
        end

        function [ what , resnorm ] = hip_lsqlin_load( obj , Cl , dl , whatpast , pin_last_zero )
            % the lsqlin call of the estimators (Kmpc.m:1354, :1442): min |C x - d|^2, x = [1;w], x_1 = 1, -1 <= w <= 1
            % and, with a previous estimate, |w_i - whatpast_i| <= 0.01 - a strictly convex QP in the free loads, solved on
            % the device (kp_qp_solve: min 1/2 x'Hx + f'x, A x <= b)
            nw = obj.params.nw;
            free = 1 : nw;
            if pin_last_zero && nw >= 1
                free = 1 : nw - 1;
            end
            what = zeros( nw , 1 );
            if ~isempty( free )
                Cf = Cl( : , 1 + free );
                r = dl - Cl( : , 1 );
                nf = numel( free );
                lo = -ones( nf , 1 ); hi = ones( nf , 1 );
                if ~isempty( whatpast )
                    wp = whatpast( end , free )';
                    lo = max( lo , wp - 0.01 ); hi = min( hi , wp + 0.01 );
                end
                Aq = [ eye(nf) ; -eye(nf) ]; bq = [ hi ; -lo ];
                what( free ) = kp_mex( 'qp_solve' , obj.hip.sys.hip.ctx , 2 * (Cf' * Cf) , -2 * (Cf' * r) , Aq , bq );
            end
            res = Cl * [ 1 ; what ] - dl;
            resnorm = res' * res;
        end

        function [ what , resnorm ] = estimate_load_linear( obj , ypast , upast , whatpast )
            % Kmpc.m:1298-1356.  The shipped code pins the LAST load to zero through the debugging equality
            % Aeq = blkdiag(1,0,1) (:1350), which only has the right size for nw = 2; reproduced for nw = 2
            if size( upast , 1 ) ~= size( ypast , 1 )
                error( 'Input arguments must have the same number of rows' );
            end
            if nargin < 4, whatpast = []; end
            traj.y = ypast; traj.u = upast;
            [ ~ , zp ] = obj.get_zeta( traj );
            hor = size( zp , 1 ); nz = obj.params.nzeta; nw = obj.params.nw; nd = obj.params.nd;
            G = obj.lift.econ_full( zp( 1:hor-1 , : )' )';        % psi of every past state, ONE device call (rows = states)
            CA = obj.model.A( 1:nz , : ); CB = obj.model.B( 1:nz , : );
            Cl = zeros( nz * (hor-1) , nw + 1 ); dl = zeros( nz * (hor-1) , 1 );
            for i = 1 : hor-1                                   % :1320-1333
                rows = nz*(i-1)+1 : nz*i;
                Cl( rows , : ) = CA * kron( eye(nw+1) , G(i,:)' );
                dl( rows ) = zp( i+1 , 1:nz )' - CB * upast( nd+i , : )';
            end
            [ what , resnorm ] = obj.hip_lsqlin_load( Cl , dl , whatpast , nw == 2 );
        end

        function [ what , resnorm ] = estimate_load_bilinear( obj , ypast , upast , whatpast )
            % Kmpc.m:1360-1444
            if size( upast , 1 ) ~= size( ypast , 1 )
                error( 'Input arguments must have the same number of rows' );
            end
            if nargin < 4, whatpast = []; end
            traj.y = ypast; traj.u = upast;
            [ ~ , zp ] = obj.get_zeta( traj );
            hor = size( zp , 1 ); nz = obj.params.nzeta; nw = obj.params.nw; m = obj.params.m;
            NL = obj.params.N * ( nw + 1 );
            G = obj.lift.econ_full( zp( 1:hor-1 , : )' )';
            Cl = zeros( nz * (hor-1) , nw + 1 ); dl = zeros( nz * (hor-1) , 1 );
            for i = 1 : hor-1                                   % :1384-1394
                Om = kron( eye(nw+1) , G(i,:)' );
                Ai = obj.model.A( 1:nz , : );
                for j = 1 : m
                    Ai = Ai + upast(i,j) * obj.model.B( 1:nz , (j-1)*NL+1 : j*NL );
                end
                rows = nz*(i-1)+1 : nz*i;
                Cl( rows , : ) = Ai * Om;
                dl( rows ) = zp( i+1 , 1:nz )';
            end
            [ what , resnorm ] = obj.hip_lsqlin_load( Cl , dl , whatpast , false );
        end

        function delete_hip( obj )
            % explicit, early release (the KpOwner member does it when the last copy of the object is gone)
            if isfield( obj.hip , 'mpc_owner' )
                obj.hip.mpc_owner.release();
            end
        end
    end

    methods ( Static )
        function obj = loadobj( s )
            % a loaded controller carries no device handle (KpOwner's handles are Transient): it is rebuilt on first use by
            % constructing the controller again from its sysid class, KmpcHip( sysid_class , ... ) - the MPC problem on the
            % device is a function of the model AND of this session's context, which a MAT file cannot hold.  Until then the
            % device calls of this object fail with kp_mex's "unknown or stale handle" error instead of touching freed memory.
            obj = s;
            if isa( obj , 'KmpcHip' ) && isstruct( obj.hip )
                obj.hip.mpc = uint64( 0 );
            end
        end
    end
end
